"""Lab tool: registers, spills, scratch and LDS of the inflate kernels of a library (default: the product).  Usage:
    python tests/tools/kernel_notes.py [path/to/libpzg.so ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
llvm = "/opt/rocm/lib/llvm/bin"
for lib in (sys.argv[1:] or [os.path.join(ROOT, "pure_zlib_amd", "libpzg.so")]):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "co.elf")
        subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "unused.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        notes = ""
        for n, at in enumerate(starts):  # one bundle per translation unit that holds kernels
            part = os.path.join(d, "fat%d.bin" % n)
            open(part, "wb").write(blob[at:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   "--input=" + part, "--output=" + co])
            notes += subprocess.check_output([llvm + "/llvm-readelf", "--notes", co]).decode()
    print(os.path.basename(lib))
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        k = {a: int(b) for a, b in re.findall(r"\.(\w+):\s+(\d+)\s*$", block, flags=re.M)}
        if "inflate" in name:
            short = re.sub(r"_ZN3pzg|EEvNS_\w+E$", "", name)
            print(f"  {short:44s} vgpr {k['vgpr_count']:3d} sgpr {k['sgpr_count']:3d} sspill {k['sgpr_spill_count']:3d} vspill {k['vgpr_spill_count']:2d} scratch {k['private_segment_fixed_size']:3d} lds {k['group_segment_fixed_size']}")
