"""FETCH_SIZE / WRITE_SIZE passes of tools/final_prof.sh -> profiles/<round>_traffic.json (what bench.py reports as roofline.traffic).

    python tools/make_traffic.py gpurun_out/r02_final_pmc_fetch.txt gpurun_out/r02_final_pmc_write.txt profiles/r02_traffic.json

Keys are the names bench.py's KernelTimer uses.  Counters are in KB (rocprofv3 derived metrics).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
counts a 128-byte request of a wide streaming read as 64 bytes and leaves other access widths to be calibrated on a known byte count in the kernel's
own pattern.  Calibrated (tools/probes/fetch_calib_probe.hip, profiles/r04_fetch_calib.txt: 1 GiB read exactly once per launch): dword loads, dwordx4
loads and LDS-DMA in rows of 128 B - 1 KB all report 0.500 of the true bytes, rows shifted by a halo pixel 0.53 (the shared lines), only isolated
64-byte rows report 1.000 — the counter tallies every request at 64 B, a full 128-byte line included.  Every entry therefore gets the x2 correction
(exact for the chain kernels' rows; an upper bound, by at most the half-line requests of the halo columns, for the masked kernels' 34-float rows — until
this calibration those were reported uncorrected, i.e. too low).  WRITE_SIZE needs none (it equals the output bytes of every kernel here).  Infinity
Cache hits are counted: this is fabric traffic of the L2s, not DRAM traffic."""
import json
import re
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+calls=\s*(\d+)\s+avg=\s*([\d.]+)", line)
            if m and cur:
                out.setdefault(cur, {})[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out


# bench.py key -> (substrings the kernel name must contain, FETCH_SIZE correction, note, algorithmic bytes per launch at batch 4 or None)
KEYS = {
    "region_modconv_sb_kernel<4,1,1,8,5>": (["region_modconv_sb_kernel<4, 1, 1, 8, 5"], 2.0,
                                            "7 launches per step (6 plain + 1 with fused ToRGB and split-plane output; on the benchmark maps the 128->256 up "
                                            "layer's launch leaves at once: its blocks are all region-uniform and run in masked_up_block_kernel); dword activation "
                                            "loads: no FETCH correction", (823900000 - 4 * (256 * 128 * 128 * 4 + 128 * 256 * 256 * 4)) // 7),
    "region_modconv_mx_kernel<1>": (["region_modconv_mx_kernel<1|region_upconv_mx4_kernel"], 2.0,
                                    "7 launches per step (the DMA-fed masked kernel, f16 + 2 x MX fp6 — six launches of region_modconv_mx_kernel and, for the 512 -> 256 @64 up "
                                    "layer, one of region_upconv_mx4_kernel, which runs the same tile code or its four-parity form per workgroup; on the benchmark maps the "
                                    "128->256 up layer's launch leaves at once); dword activation loads, 16-byte weight DMA: FETCH_SIZE x2 (calibrated; an upper bound by the halo columns' half-line requests)",
                                    (823900000 - 4 * (256 * 128 * 128 * 4 + 128 * 256 * 256 * 4)) // 7),
    "region_modconv_mx_kernel<0>": (["region_modconv_mx_kernel<0"], 2.0, "as <1> with the split-bf16 arithmetic", (823900000 - 4 * (256 * 128 * 128 * 4 + 128 * 256 * 256 * 4)) // 7),
    "masked_upconv_blocks": (["masked_up_block_mx_kernel"], 2.0, "1 launch per step (the 256 -> 128 @128 masked up layer: every 16 x 16 output block of the benchmark maps lies under one "
                             "region); algorithmic bytes = its input [4,256,128,128] + its output [4,128,256,256], fp32; dword loads, FETCH_SIZE x2 (calibrated)",
                             4 * (256 * 128 * 128 * 4 + 128 * 256 * 256 * 4)),
    "chain_conv3x3<32>": (["chain_conv_kernel<1, 2"], 2.0, "LDS-DMA dwordx4 only: FETCH_SIZE x2 (guide)", 4 * (32 * 1024 * 1024 * 4 + 3 * 1024 * 1024 * 4 + 3 * 512 * 512 * 4)),
    "chain_conv3x3<64>": (["chain_conv_kernel<2, 4"], 2.0, "LDS-DMA dwordx4 only: FETCH_SIZE x2 (guide)", 4 * (2 * 64 * 512 * 512 * 4 + 3 * 512 * 512 * 4 + 3 * 256 * 256 * 4)),
    "modconv_up_hc": (["up_hcp_kernel"], 2.0, "2 launches per step (256->512, 512->1024), half-composed form, split-plane in/out, LDS-DMA dwordx4 only: FETCH_SIZE x2 (guide)",
                      (4 * (128 * 256 * 256 * 4 + 64 * 512 * 512 * 4) + 4 * (64 * 512 * 512 * 4 + 32 * 1024 * 1024 * 4)) // 2),
    "modconv_up_fused_sb": (["up_fused"], 2.0, "2 launches per step (256->512, 512->1024), split-plane in/out, 16-byte loads: FETCH_SIZE x2 (guide)",
                            (4 * (128 * 256 * 256 * 4 + 64 * 512 * 512 * 4) + 4 * (64 * 512 * 512 * 4 + 32 * 1024 * 1024 * 4)) // 2),
}


def main(fetch_path, write_path, out_path):
    f, w = parse(fetch_path), parse(write_path)
    doc = {}
    for key, (subs, corr, note, alg) in KEYS.items():
        hit = lambda n: all(any(a in n for a in s.split("|")) for s in subs)      # ("a|b": either name)
        fk = [(n, v["FETCH_SIZE"]) for n, v in f.items() if hit(n) and "FETCH_SIZE" in v]
        wk = [(n, v["WRITE_SIZE"]) for n, v in w.items() if hit(n) and "WRITE_SIZE" in v]
        if not fk or not wk:
            continue
        nf, nw = sum(c for _, (c, _) in fk), sum(c for _, (c, _) in wk)
        fetch_kb = sum(c * a for _, (c, a) in fk) / nf
        write_kb = sum(c * a for _, (c, a) in wk) / nw
        doc[key] = {"fetch_kb_per_launch_raw": round(fetch_kb, 1), "fetch_correction": corr, "write_kb_per_launch": round(write_kb, 1),
                    "hbm_bytes_per_launch": int((fetch_kb * corr + write_kb) * 1024), "launches_averaged": nf, "algorithmic_bytes_per_launch": alg,
                    "note": note,
                    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python3 bench.py --steps 2 --warmup 1 "
                              f"--no-cpu-baseline --no-full-swap --no-pti --clip 0` (tools/final_prof.sh -> {fetch_path.split('/')[-1]}, {write_path.split('/')[-1]})"}
    with open(out_path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: (v["hbm_bytes_per_launch"], v["algorithmic_bytes_per_launch"]) for k, v in doc.items()}))


if __name__ == "__main__":
    main(*sys.argv[1:4])
