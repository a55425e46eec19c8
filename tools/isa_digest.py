"""sha1 of the machine code of every gfx950 kernel of a HIP source (device-only compile with the library's flags, bytes of each function
from the code object's symbol table).  Used to show that a source change left existing kernels' code untouched:

    python tools/isa_digest.py action-detection_amd/csrc/conv_pl.hip > /tmp/before.txt;  ...edit...;  same > /tmp/after.txt;  diff
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def digests(src, inc):
    with tempfile.TemporaryDirectory() as d:
        co, elf = os.path.join(d, "a.co"), os.path.join(d, "a.elf")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src, "-o", co, "-I", inc])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + co,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf, "--unbundle"])
        syms = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", elf], capture_output=True, text=True).stdout
        secs = subprocess.run([LLVM + "/llvm-readelf", "-S", "-W", elf], capture_output=True, text=True).stdout
        m = re.search(r"\]\s+\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
        addr, off = int(m.group(1), 16), int(m.group(2), 16)
        blob = open(elf, "rb").read()
        out = {}
        for line in syms.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC" and f[6] != "UND":
                a, size, name = int(f[1], 16), int(f[2]), f[7]
                code = blob[off + a - addr: off + a - addr + size]
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                out[re.sub(r"\(anonymous namespace\)::", "", dem)] = (size, hashlib.sha1(code).hexdigest()[:16])
        return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    inc = os.path.join(here, "..", "action-detection_amd", "csrc")
    for s in sys.argv[1:]:
        for name, (size, h) in sorted(digests(s, inc).items()):
            print("%s %6d %s" % (h, size, name))
