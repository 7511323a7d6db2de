#!/usr/bin/env python3
"""kernel_resources.py LIB.so [--check] -- what every gfx950 kernel inside the library asks of the machine, read from the code objects
themselves (llvm-objdump --offloading + llvm-readelf --notes): VGPRs (callees included: the assembler takes the maximum over the call
graph), waves per SIMD that allocation allows, spilled registers, private (scratch) bytes per lane, static LDS, workgroup size.

--check (run by `make hip`) fails the build when
  * a kernel's register allocation cannot host the workgroup its launch bounds promise (VGPRs rounded up to 8, 512 per SIMD), or
  * one of the state-machine megakernels (mega_bvh_kernel, mega_gen_kernel) touches private memory beyond a callee frame or spills
    a vector register: their state loops are built to run without scratch.  The 1024-thread variants are compiled
    for 128 VGPRs (four waves per SIMD): mega_bvh_kernel<1024> may keep at most 32 registers / 128 B in private memory (rarely
    touched per-pixel values, read and written in the shade step only), mega_gen_kernel<1024> is a measurement build.
"""
import os, re, subprocess, sys, tempfile, shutil
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
lib = sys.argv[1]
check = "--check" in sys.argv
tmp = tempfile.mkdtemp(prefix="mortres")
try:
    work = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, work)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], check=True, capture_output=True)
    rows = []
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)
            name = g("name").group(1)
            filt = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
            dem = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if os.path.exists(filt) else name
            rows.append(dict(name=dem.replace("void ", "").split("(")[0], vgpr=int(g("vgpr_count").group(1)), sgpr=int(g("sgpr_count").group(1)),
                             vspill=int(g("vgpr_spill_count").group(1)), sspill=int(g("sgpr_spill_count").group(1)),
                             private=int(g("private_segment_fixed_size").group(1)), lds=int(g("group_segment_fixed_size").group(1)),
                             wg=int(g("max_flat_workgroup_size").group(1))))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
bad = []
print(f"{'kernel':58s} {'wg':>5s} {'vgpr':>5s} {'waves/SIMD':>10s} {'v-spill':>7s} {'s-spill':>7s} {'private B':>9s} {'static LDS':>10s}")
OWN = ("mega_", "wf_", "seed_kernel", "calib_", "deinterleave", "substream", "iota_kernel")
for r in sorted(rows, key=lambda r: r["name"]):
    if not any(o in r["name"] for o in OWN) and "--all" not in sys.argv:
        continue  # rocPRIM's sort kernels (tile_sort.hip): listed with --all
    alloc = (r["vgpr"] + 7) // 8 * 8
    waves = min(8, 512 // max(alloc, 8))
    need = (r["wg"] + 255) // 256  # waves per SIMD one workgroup brings
    print(f"{r['name'][:58]:58s} {r['wg']:5d} {r['vgpr']:5d} {waves:10d} {r['vspill']:7d} {r['sspill']:7d} {r['private']:9d} {r['lds']:10d}")
    if need > waves:
        bad.append(f"{r['name']}: {r['vgpr']} VGPRs allow {waves} waves per SIMD, a {r['wg']}-thread workgroup needs {need}")
    if ("mega_bvh_kernel" in r["name"] or "mega_gen_kernel" in r["name"]) and not r["name"].startswith("mega_gen_kernel<1024"):
        wide = r["name"].startswith("mega_bvh_kernel<1024")
        if (r["vspill"] > (32 if wide else 0)) or r["private"] > 128:
            bad.append(f"{r['name']}: {r['vspill']} spilled VGPRs, {r['private']} B of private memory per lane (state loop must run without scratch)")
if bad:
    print("\n".join("RESOURCE CHECK FAILED: " + b for b in bad), file=sys.stderr)
    sys.exit(1 if check else 0)
