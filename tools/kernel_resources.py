"""Register / LDS / scratch usage of every kernel in libpyqmc_amd.so (from the code object's metadata notes).
usage: python tools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("PQA_LIB", os.path.join(ROOT, "pyqmc_amd", "lib", "libpyqmc_amd.so"))
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    raw = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, raw)]  # one bundle per translation unit of the library
    notes = ""
    for i, s in enumerate(starts):
        one, co = os.path.join(d, f"b{i}.bin"), os.path.join(d, f"dev{i}.co")
        open(one, "wb").write(raw[s : starts[i + 1] if i + 1 < len(starts) else len(raw)])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={one}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
rows = []
for k in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
    k = ".agpr_count:" + k
    g = lambda key: (re.search(r"\." + key + r":\s*(\S+)", k) or [None, "?"])[1]
    rows.append([g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")])
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    if len(sys.argv) > 1 and not any(s in n for s in sys.argv[1:]):
        continue
    tot = int(r[1]) if r[1].isdigit() else 0  # .vgpr_count is the unified total (arch + acc) on gfx90a+
    alloc = -(-tot // 8) * 8
    print(f"{n[:90]:90s} vgpr {r[1]:>3s} agpr {r[2]:>3s} ({512 // max(alloc, 1) if alloc else '?'} waves/SIMD) sgpr {r[3]:>3s} lds {r[4]:>6s} scratch {r[5]}")
