#!/usr/bin/env python
"""Registers / scratch / LDS of every kernel in a built libbrutus_amd.so, read from the
AMDGPU metadata of the gfx950 code object inside it (no recompilation):

    python tools/kernel_resources.py [lib.so] [name substring ...]

A hot kernel that starts to spill (private segment > 0) or crosses a VGPR occupancy step
does so silently -- round 4 lost 25 % of k_fflux that way for one commit.
tests/test_cabi.py::test_hot_kernels_do_not_spill runs this on every build."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(so):
    """The gfx950 code objects of the library: one offload bundle per translation unit, laid
    end to end (page-aligned) in .hip_fatbin."""
    out = subprocess.check_output([READELF, "-S", "-W", so], text=True)
    m = re.search(r"\.hip_fatbin\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", out)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    with open(so, "rb") as f:
        f.seek(off)
        data = f.read(size)
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    assert data[:24] == magic, "compressed / unknown offload bundle"
    found, at = [], data.find(magic)
    while at >= 0:
        n = struct.unpack_from("<Q", data, at + 24)[0]
        p = at + 32
        for _ in range(n):
            o, s, t = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + t].decode()
            p += 24 + t
            if "gfx950" in triple:
                found.append(data[at + o:at + o + s])
        at = data.find(magic, at + 24)
    if not found:
        raise RuntimeError("no gfx950 code object in %s" % so)
    return found


def kernels(so):
    """{demangled kernel name: dict(vgpr, sgpr, scratch, lds)}"""
    notes = ""
    for co in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes += subprocess.check_output([READELF, "--notes", f.name], text=True)
    res, cur = {}, {}
    keys = {".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".private_segment_fixed_size": "scratch",
            ".group_segment_fixed_size": "lds", ".name": "name", ".vgpr_spill_count": "vgpr_spills"}
    for line in notes.splitlines():
        m = re.match(r"\s*(?:-\s+)?(\.[a-z_]+):\s+(.*)$", line)
        if not m or m.group(1) not in keys:
            continue
        k, v = keys[m.group(1)], m.group(2).strip().strip("'\"")
        if k in cur and k == "name" or (k != "name" and k in cur and "name" in cur and len(cur) >= 6):
            pass
        cur[k] = v
        if len(cur) == len(keys):
            res[cur["name"]] = cur
            cur = {}
    names = list(res)
    dem = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                         capture_output=True).stdout.splitlines()
    out = {}
    for n, d in zip(names, dem):
        d = d.replace("(anonymous namespace)::", "")
        d = re.sub(r"^void ", "", d).split("(")[0]
        out[d] = {k: int(v) for k, v in res[n].items() if k != "name"}
    return out


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else \
        os.path.join(ROOT, "brutus_amd", "libbrutus_amd.so")
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for name, r in sorted(kernels(so).items()):
        if not pats or any(p in name for p in pats):
            print("%-58s vgpr %3d  sgpr %3d  scratch %4d  lds %6d" % (name[:58], r["vgpr"], r["sgpr"],
                                                                     r["scratch"], r["lds"]))
