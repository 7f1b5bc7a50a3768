#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh (rocpd databases):
HBM-side bytes per launch of the kernels bench.py reports rooflines for, corrected as MI355X_MICROARCH.md prescribes
(FETCH_SIZE is counted in KB of 32-byte requests on gfx950 for 16-byte-per-lane reads: doubled; WRITE_SIZE in KB taken as is).
usage: python tools/pmc_to_traffic.py FETCH_DIR WRITE_DIR > profiles/r02_traffic.json"""
import glob, json, os, sqlite3, sys
from collections import defaultdict

UNITS = {  # kernel name fragment -> (key in the JSON, units per launch key, units per launch of bench.py's default shapes)
    "tdec_win_kernel<8, phyhip::turbo::Ar16, false>": ("tdec_win_kernel", "code_blocks_per_launch", 131040),
    "ofdm_kernel<phyhip::fft::Plan<2048": ("ofdm_kernel", "subframes_per_launch", 10080),
    "ldpc_packed_kernel<false, false>": ("ldpc_packed_kernel", "code_words_per_launch", 16384),
    "pss_wave_kernel": ("pss_wave_kernel", "captures_per_launch", 256),
    "ofdm_kernel<phyhip::fft::Plan<4096": ("ofdm_kernel_n4096", "slots_per_launch", 2048),
    "tdec_win_kernel<16, phyhip::turbo::Ar8, false>": ("tdec_win_kernel_8bit", "code_blocks_per_launch", 131040),
}
# the kernels of one step of the uplink leg (extra.uplink: 64 UEs x 184 subframes): their bench-sized dispatches summed
CHAIN = {"uplink_chain": (["ofdm_kernel<phyhip::fft::Plan<1536", "modem::", "dft_fixed_kernel<phyhip::fft::Plan<1200", "rm_rx_gather_lds_kernel<short>",
                           "tdec_win_kernel<8, phyhip::turbo::Ar16, true>", "tb_crc_kernel"], "ue_subframes_per_step", 64 * 184)}


def mean_counter(d, counter):
    acc = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
        cur = sqlite3.connect(path).cursor()
        for n, v in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            acc[n].append(v)
    return {n: max(v) for n, v in acc.items()}  # the bench-sized dispatches are the largest ones of each kernel


def main():
    fetch, write = mean_counter(sys.argv[1], "FETCH_SIZE"), mean_counter(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) on `python bench.py --steps 2 --warmup 1 "
                     "--extra-steps 2 --no-cpu`; FETCH_SIZE (KB) doubled for 16-byte-per-lane reads on gfx950 as MI355X_MICROARCH.md prescribes, "
                     "WRITE_SIZE (KB) taken as is; largest dispatch of each kernel (= the bench-sized launch)"}
    for frag, (key, ukey, units) in UNITS.items():
        f = [v for n, v in fetch.items() if frag in n]
        w = [v for n, v in write.items() if frag in n]
        if not f or not w:
            continue
        fk, wk = max(f), max(w)  # the bench-sized dispatches (set-up dispatches of the same kernel are smaller)
        out[key] = {"fetch_size_kb": fk, "write_size_kb": wk, "traffic_bytes_per_launch": (2 * fk + wk) * 1024, ukey: units}
    for key, (frags, ukey, units) in CHAIN.items():
        tot, parts = 0.0, {}
        for frag in frags:
            for n in sorted(set(fetch) | set(write)):
                if frag in n and n in fetch and n in write:
                    b = (2 * fetch[n] + write[n]) * 1024
                    parts[n[:90]] = b
                    tot += b
        if parts:
            out[key] = {"traffic_bytes_per_launch": tot, ukey: units, "kernels": parts}
    print(json.dumps(out, indent=1))


main()
