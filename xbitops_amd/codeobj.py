"""Kernel resource usage read from the built library itself (diagnostics; not on the compute path).

`hipcc` embeds one clang offload bundle per translation unit in the shared library; each holds the gfx950
code object, whose NT_AMDGPU_METADATA note (msgpack) lists every kernel with its register and LDS use.
``kernel_resources()`` returns {mangled kernel name: {"vgpr", "sgpr", "spill", "scratch", "lds"}}.
Used by tools/kernel_resources.py and by the test that pins the split-K planner's occupancy table
(kSplitVgprs in csrc/xbit_gemv_mfma.h) to the library that actually ships.
"""
from __future__ import annotations

import re
import struct

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_GEMV = re.compile(r"gemv_mfma_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELb(\d)E")
GEMV_FIELDS = ("bits", "ub", "cpl", "nl", "nw", "mt", "nt", "guard", "ablate", "la", "split")


def _notes(elf: bytes):
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    for s in range(shnum):
        _name, typ, _flags, _addr, offset, size = struct.unpack_from("<IIQQQQ", elf, shoff + s * shentsize)
        if typ != 7:  # SHT_NOTE
            continue
        q = offset
        while q + 12 <= offset + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, q)
            q += 12 + ((namesz + 3) & ~3)
            yield ntype, elf[q:q + descsz]
            q += (descsz + 3) & ~3


def kernel_resources(path: str, arch: str = "gfx950") -> dict:
    import msgpack

    data = open(path, "rb").read()
    out = {}
    for m in re.finditer(_MAGIC, data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        off = p + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if arch not in triple or sz == 0:
                continue
            for ntype, desc in _notes(data[p + o:p + o + sz]):
                if ntype != 32:  # NT_AMDGPU_METADATA
                    continue
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = {"vgpr": k[".vgpr_count"], "sgpr": k[".sgpr_count"], "spill": k.get(".vgpr_spill_count", 0),
                                       "scratch": k.get(".private_segment_fixed_size", 0), "lds": k.get(".group_segment_fixed_size", 0)}
    return out


def gemv_template_args(mangled: str):
    """Template arguments of a gemv_mfma_kernel instantiation as a dict (GEMV_FIELDS), or None."""
    m = _GEMV.search(mangled)
    return dict(zip(GEMV_FIELDS, (int(x) for x in m.groups()))) if m else None
