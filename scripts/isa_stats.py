"""VGPR / scratch / LDS of the kernels in the built library (from the code-object metadata): python scripts/isa_stats.py [substr ...]
Reads sadvio_amd/csrc/libsadvio_ba.so with /opt/rocm/lib/llvm/bin/llvm-readelf --notes (amdhsa.kernels)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("SADVIO_BA_LIB", os.path.join(ROOT, "sadvio_amd", "csrc", "libsadvio_ba.so"))
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={lib}",
                f"--output={tmp}/k.co", "--unbundle"], check=False, capture_output=True)
co = f"{tmp}/k.co"
if not os.path.exists(co) or os.path.getsize(co) == 0:
    # the fat binary sits in the .hip_fatbin section of the host library
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{tmp}/fat.bin"], check=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={tmp}/fat.bin",
                    f"--output={co}", "--unbundle"], check=True)
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
pats = sys.argv[1:]
for blk in txt.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pats and not any(p in dem for p in pats):
        continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk)
    print(dem[:90].ljust(92), "vgpr", g("vgpr_count").group(1), "spill", g("vgpr_spill_count").group(1), "scratch", g("private_segment_fixed_size").group(1),
          "lds", g("group_segment_fixed_size").group(1))
