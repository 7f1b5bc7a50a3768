#!/usr/bin/env python3
"""The reference's OWN test programs as a stopwatch: its unmodified pdsch_test / pusch_test print the time srsran_pdsch_decode /
srsran_pusch_decode took (pdsch_test.c:480-499, pusch_test.c:326-396).  Three link sets of the same program (tests/ref_link/Makefile):

  refcpu  the reference's own objects on the host cores (everything but its three FFT files): the CPU baseline by the reference itself
  full    the library bound per code block (srsran_rm_turbo_rx_lut + srsran_tdec_iteration + CRC per block,
          a device round trip per half iteration)
  tb      the library bound at the reference's transport-block seam decode_tb_cb (sch.c:370; tests/ref_link/tb_bind.c): one call per block
  chan    the library bound at the grant level (tests/ref_link/chan_bind.c: srsran_pusch_decode / srsran_pdsch_decode / srsran_pdsch_encode /
          srsran_ulsch_encode): ONE device call per grant, everything between the resource grid and the transport block resident

Run on the GPU box:  python tools/bench_ref_programs.py > gpurun_out/ref_programs.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "ref_link", "_build")

CASES = [
    # (program, args, label).  pusch_test decodes a fresh subframe per iteration; pdsch_test -X n decodes ONCE and skips the n - 1 repetitions
    # (pdsch.c:893: a block whose crc flag is set is not decoded again), so only its single, cold decode is a measurement
    ("pusch_test", ["-n", "100", "-L", "100", "-m", "28", "-p", "enable_64qam", "-s", "40"], "PUSCH 100 PRB, MCS 28, 64-QAM (TBS 75376: 13 code blocks), 40 subframes"),
    ("pusch_test", ["-n", "100", "-L", "50", "-m", "21", "-p", "uci_ack", "2", "-p", "cqi", "wideband", "-s", "40"], "PUSCH 50 of 100 PRB, MCS 21, ACK + CQI multiplexed"),
    ("pusch_test", ["-n", "100", "-L", "50", "-m", "21", "-s", "40"], "PUSCH 50 of 100 PRB, MCS 21"),
    ("pusch_test", ["-n", "25", "-L", "25", "-m", "14", "-s", "40"], "PUSCH 25 PRB, MCS 14"),
    ("pusch_test", ["-n", "15", "-L", "12", "-m", "14", "-s", "40"], "PUSCH 12 of 15 PRB, MCS 14"),
    ("pusch_test", ["-n", "6", "-L", "6", "-m", "0", "-s", "40"], "PUSCH 6 PRB, MCS 0 (one small code block)"),
    ("pdsch_test", ["-n", "100", "-m", "28", "-X", "1"], "PDSCH 100 PRB, MCS 28: ONE cold decode (first call of the process)"),
    # the transmit side: pdsch_test encodes the block -X times for real (pdsch_test.c:421-436) -- us_per_encode; its decode figure of such a run is void (above)
    ("pdsch_test", ["-n", "100", "-m", "28", "-X", "50"], "PDSCH 100 PRB, MCS 28: srsran_pdsch_encode x 50 (us_per_encode; the decode figure of this run is void)"),
    ("pdsch_test", ["-n", "25", "-m", "20", "-X", "50"], "PDSCH 25 PRB, MCS 20: srsran_pdsch_encode x 50 (us_per_encode)"),
    ("pdsch_test", ["-n", "6", "-m", "10", "-X", "50"], "PDSCH 6 PRB, MCS 10: srsran_pdsch_encode x 50 (us_per_encode)"),
    # the eNB-DL -> UE-DL loop-back harness of BASELINE configs[0]: whole subframes (its own Mbit/s figures: bits over the summed encode / decode times)
    ("phy_dl_test", ["-p", "6", "-t", "1", "-m", "28"], "phy_dl_test 6 PRB, TM1, MCS 28 (configs[0]'s cell)"),
    ("phy_dl_test", ["-p", "100", "-t", "1", "-m", "28"], "phy_dl_test 100 PRB, TM1, MCS 28 (configs[1]'s cell)"),
]


def run(kind, prog, args):
    exe = os.path.join(BUILD, kind, prog)
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([exe] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, errors="replace", timeout=600)
    out = r.stdout
    res = {"rc": r.returncode}
    m = re.search(r"DECODED (\w+) in ([0-9.]+) \(PHY bitrate=([0-9.]+) Mbps\. Processing bitrate=([0-9.]+) Mbps\)", out)
    if m:  # pdsch_test: microseconds per srsran_pdsch_decode, averaged by the program over its -X repetitions
        res.update(ok=m.group(1) == "OK", us_per_decode=float(m.group(2)), mbps=float(m.group(4)))
    m = re.search(r"ENCODED in ([0-9.]+) \(PHY bitrate=([0-9.]+) Mbps\. Processing bitrate=([0-9.]+) Mbps\)", out)
    if m:
        res.update(us_per_encode=float(m.group(1)))
    m = re.search(r"Decoded Rate: ([0-9.]+) Mbps", out)
    if m:  # pusch_test: bits over the summed decode times of its subframes; the first subframe of a process pays the one-time set-up
        per = [float(x) for x in re.findall(r"Processing: ([0-9.]+) Mbps", out)]
        tbs = re.search(r"TBS: (\d+) bits", out)
        res.update(mbps=float(m.group(1)), mbps_steady=sorted(per)[len(per) // 2] if per else None,
                   us_per_decode_steady=(int(tbs.group(1)) / sorted(per)[len(per) // 2]) if per and tbs else None)
    m = re.search(r"eNb:\s+([0-9.]+)\s+([0-9.]+)\s+UE:\s+([0-9.]+)\s+([0-9.]+)", out)
    if m:  # phy_dl_test: "Processed" = received bits per microsecond spent in the eNB's encode / the UE's decode (whole subframe, OFDM and channel estimation included)
        res.update(ok=r.returncode == 0, enb_processed_mbps=float(m.group(2)), ue_processed_mbps=float(m.group(4)), mbps=float(m.group(4)))
    if "mbps" not in res:
        res["tail"] = out[-400:]
    return res


def main():
    rows = []
    for prog, args, label in CASES:
        row = {"program": prog, "args": " ".join(args), "what": label}
        for kind, key in (("bin_refcpu", "reference_cpu_1_core"), ("bin_full", "library_per_code_block"), ("bin_tb", "library_transport_block_seam"),
                          ("bin_chan", "library_grant_seam")):
            row[key] = run(kind, prog, args)
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps({"host_cpus": os.cpu_count(), "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
