"""Copies one tools/profile_all.sh run (gpurun_out/<tag>/) into profiles/ as <tag>-named files and refreshes
profiles/pmc_latest.json (which bench.py reads for roofline.traffic).   usage: python tools/store_evidence.py r01_v12 "note" """
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; note = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", tag); dst = os.path.join(ROOT, "profiles")
rnd, ver = tag.split("_", 1)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{rnd}_bench_{ver}.json"))
shutil.copy(os.path.join(src, "kernel_stats.md"), os.path.join(dst, f"{rnd}_kernel_stats_{ver}.md"))
t = open(os.path.join(src, "pmc_k_packet.txt")).read()
v = {l.split()[0]: float(l.split()[1]) for l in t.strip().splitlines() if len(l.split()) >= 2}
W = v["SQ_WAVES"]; fetch = v["FETCH_SIZE"] * 1024; write = v["WRITE_SIZE"] * 1024
cycles = v["GRBM_GUI_ACTIVE"] / 8; cap = cycles * 1024 / 2
alg = 538670936
d = f"""
derived (per launch of k_packet<GRID,UNSIGNED,GRID_PLANE>, 512^3 x blob-100k; {note}):
  waves {W:.0f}; VALU/wave {v['SQ_INSTS_VALU']/W:.0f}; SALU/wave {v['SQ_INSTS_SALU']/W:.0f}; SMEM/wave {v['SQ_INSTS_SMEM']/W:.0f}; VMEM_RD/wave {v['SQ_INSTS_VMEM_RD']/W:.2f}; VMEM_WR/wave {v['SQ_INSTS_VMEM_WR']/W:.0f}; LDS {v['SQ_INSTS_LDS']:.0f}
  shader cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs) {cycles/1e6:.1f} M  -> VALU issue capacity (1024 SIMD-32, 2 cycles per wave64 op) {cap/1e9:.2f} G ops; issued {v['SQ_INSTS_VALU']/1e9:.2f} G = {v['SQ_INSTS_VALU']/cap:.3f} of the plain-op issue ceiling
  resident waves per SIMD (SQ_WAVE_CYCLES quad-cycles x4 / SIMD-cycles) {v['SQ_WAVE_CYCLES']*4/(cycles*1024):.2f} of 8
  wave time split: active {v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES']:.3f}  issue-wait {v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.3f} (overlaps)  waitcnt {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.3f}
  L2 hit rate {v['TCC_HIT_sum']/(v['TCC_HIT_sum']+v['TCC_MISS_sum']):.4f}
  FETCH_SIZE {fetch/1e6:.1f} MB as reported (x2 = {2*fetch/1e6:.1f} MB with the gfx950 half-count correction; these are L2 misses, the 26 MB of BVH records
  re-read by the eight XCD L2s out of the 256 MB Infinity Cache, which the counter does not exclude), WRITE_SIZE {write/1e6:.1f} MB (known byte count of the output: 536.9 MB); algorithmic {alg/1e6:.1f} MB
  memory-side traffic per launch = {(2*fetch+write)/1e6:.1f} MB = {(2*fetch+write)/alg:.3f} x algorithmic ({(fetch+write)/alg:.3f} x with FETCH_SIZE as reported)
"""
open(os.path.join(dst, f"{rnd}_pmc_k_packet_{ver}.txt"), "w").write(t + d)
if os.path.exists(os.path.join(src, "pmc_k_cut.txt")):
    shutil.copy(os.path.join(src, "pmc_k_cut.txt"), os.path.join(dst, f"{rnd}_pmc_k_cut_{ver}.txt"))
json.dump({"k_packet_hbm_bytes_per_launch": 2 * fetch + write, "fetch_bytes_as_reported": fetch, "write_bytes": write,
           "valu_issue_frac": round(v["SQ_INSTS_VALU"] / cap, 3), "source": f"profiles/{rnd}_pmc_k_packet_{ver}.txt",
           "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_all.sh), KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 half-count; an upper estimate for our 16-96 B broadcast / scalar loads; the counter includes Infinity-Cache hits); WRITE_SIZE checked against the known 536.9 MB output; k_packet {ver}, 512^3 x blob-100k; source profiles/{rnd}_pmc_k_packet_{ver}.txt. valu_issue_frac = SQ_INSTS_VALU / (1024 SIMD-32 x GRBM_GUI_ACTIVE/8 / 2 cycles per wave64 op)"},
          open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
sec = os.path.join(dst, f"{rnd}_secondary_{ver}.txt")
with open(sec, "w") as f:
    f.write("# secondary measurements, MI355X (tools/exp_configs.py, exp_host_calls.py, exp_slabs.py, exp_piece_size.py)\n")
    for title, name in (("other BASELINE configs, device resident", "configs.txt"),
                        ("drop-in (host pointer) calls, numpy in / numpy out, wall time", "host_calls.txt"),
                        ("one rank's work of an N-GPU, 4-chunk step, run alone on one GPU (no collectives; pieces on two alternating streams)", "slabs.txt"),
                        ("the 512^3 grid computed in x-pieces of L layers, one after the other on one stream: sums of the per-piece phase times", "piece_size.txt"),
                        ("low-poly mesh (suzanne, 968 triangles): large and small grids", "suzanne.txt")):
        pth = os.path.join(ROOT, "gpurun_out", name)
        if os.path.exists(pth):
            f.write(f"## {title}\n" + open(pth).read())
print(d); print(open(sec).read())
