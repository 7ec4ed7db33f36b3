#!/usr/bin/env python
"""Register / LDS / spill figures of every gfx950 kernel in a hipcc object or shared library: pulls the code objects out of
the clang offload bundle (__CLANG_OFFLOAD_BUNDLE__ in .hip_fatbin) and prints llvm-readelf's metadata notes.

    python tools/kernel_resources.py commonscenes_amd/build/cs_gemm_f16x3.o [name-filter]
"""
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        q = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield blob[i + off:i + off + size]
        pos = i + 24


def main(path, flt=""):
    blob = open(path, "rb").read()
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count" and cur:
                pass
            if k in ("agpr_count", "group_segment_fixed_size", "name", "private_segment_fixed_size", "sgpr_count",
                     "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "symbol"):
                cur[k] = v
            if k == "wavefront_size":
                name = cur.get("name", "?")
                d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                d = re.sub(r"\(anonymous namespace\)::", "", d)
                d = re.sub(r"\(.*$", "", d)
                if flt in d:
                    print(f"{d[:90]:90s} vgpr {cur.get('vgpr_count'):>4s} agpr {cur.get('agpr_count', '0'):>4s} sgpr {cur.get('sgpr_count'):>4s} "
                          f"lds {cur.get('group_segment_fixed_size'):>7s} scratch {cur.get('private_segment_fixed_size'):>5s} "
                          f"spill v{cur.get('vgpr_spill_count')} s{cur.get('sgpr_spill_count')}")
                cur = {}


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
