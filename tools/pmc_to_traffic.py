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
    "ldpc_packed_kernel<false>": ("ldpc_packed_kernel", "code_words_per_launch", 16384),
    "pss_wave_kernel": ("pss_wave_kernel", "captures_per_launch", 256),
}


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
    print(json.dumps(out, indent=1))


main()
