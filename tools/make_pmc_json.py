"""profiles/pmc_pairs_fft.json from the two PMC summaries of `tools/gpu_visit.sh hbm` (FETCH_SIZE and WRITE_SIZE passes over the default
bench command): HBM-side bytes of one unit = pairs_fft_plane + pairs_fft_lines + pairs_fft_plane, with the hash of the kernel
sources they were measured on (bench.py quotes the figure only while that hash matches).
usage: make_pmc_json.py <fetch_summary.txt> <write_summary.txt> <tag>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def column(path, counter, kernel):
    lines = open(path).read().splitlines()
    header = lines[0].split()
    col = header.index(kernel)
    for line in lines:
        parts = line.split()
        if parts and parts[0] == counter:
            return float(parts[1 + col])
    raise KeyError((counter, kernel))


def main():
    fetch, write, tag = sys.argv[1:4]
    import bench
    f_plane, f_lines = column(fetch, "FETCH_SIZE", "pairs_fft_plane"), column(fetch, "FETCH_SIZE", "pairs_fft_lines")
    w_plane, w_lines = column(write, "WRITE_SIZE", "pairs_fft_plane"), column(write, "WRITE_SIZE", "pairs_fft_lines")
    n = column(fetch, "n", "pairs_fft_lines")
    # gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; both in KB
    traffic = 1024.0 * (2.0 * (2 * f_plane + f_lines) + (2 * w_plane + w_lines))
    out = {"kernel": "pairs_fft_plane + pairs_fft_lines + pairs_fft_plane (one unit = the three launches)",
           "source": "profiles/%s_pmc_fetch.txt and %s_pmc_write.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, %d units each)" % (tag, tag, int(n)),
           "fetch_size_kb": {"pairs_fft_plane (x2)": f_plane, "pairs_fft_lines": f_lines},
           "write_size_kb": {"pairs_fft_plane (x2)": w_plane, "pairs_fft_lines": w_lines},
           "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated, taken as reported",
           "traffic_bytes_per_launch": int(traffic), "kernel_sources_sha": bench.kernel_sources_sha(),
           "visit": "%s, %s, device %s" % (tag, __import__("datetime").date.today().isoformat(), os.environ.get("OMMHIP_VISIT_BOX", "MI355X gpurun box"))}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_pairs_fft.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
