"""Register / LDS / spill report of a HIP source's gfx950 kernels (device-only compile + llvm-readelf notes)."""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def report(src, inc):
    with tempfile.TemporaryDirectory() as d:
        co, elf = os.path.join(d, "a.co"), os.path.join(d, "a.elf")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-c", src, "-o", co, "-I", inc])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + co,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf, "--unbundle"])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        g = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, k).group(1))   # noqa: E731
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem)
        dem = re.sub(r"\((x6::X6Args|WgX6Args|ConvArgs|[A-Za-z:]*Args)\)", "", dem)
        print("%-64s vgpr %3d  spill v%-3d s%-3d  lds %6d" % (dem[:64], g("vgpr_count"), g("vgpr_spill_count"),
                                                             g("sgpr_spill_count"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    inc = os.path.join(here, "..", "action-detection_amd", "csrc")
    for s in sys.argv[1:]:
        report(s, inc)
