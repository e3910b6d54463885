#!/usr/bin/env python3
"""Per-kernel resource usage of libl3d_hip.so read from the code objects inside it (no GPU, no ROCm tools): the .hip_fatbin
section holds one clang offload bundle per translation unit; each bundle's gfx950 entry is an ELF whose AMDGPU metadata note
(msgpack) lists every kernel with its VGPR / SGPR counts, LDS and scratch (private segment) sizes and spill counts.
    python tools/kernel_meta.py [pattern]        prints the table
tests/test_host_cpu.py uses it to keep the hot kernels free of scratch: a spill there is a silent 10-30 % (LABLOG R2.4h)."""
import os
import re
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stroff = secs[shstrndx][2]
    out = {}
    for name, typ, off, size in secs:
        end = elf.index(b"\0", stroff + name)
        out.setdefault(elf[stroff + name:end].decode(), []).append((typ, off, size))
    return out


def _code_objects(so_bytes):
    """every gfx9xx ELF inside the .hip_fatbin section"""
    secs = _sections(so_bytes)
    for typ, off, size in secs.get(".hip_fatbin", []):
        blob = so_bytes[off:off + size]
        pos = blob.find(MAGIC)
        while pos >= 0:
            n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
            p = pos + len(MAGIC) + 8
            for _ in range(n):
                eoff, esize, tsize = struct.unpack_from("<QQQ", blob, p)
                triple = blob[p + 24:p + 24 + tsize].decode()
                p += 24 + tsize
                if "amdgcn" in triple and esize:
                    yield triple, blob[pos + eoff:pos + eoff + esize]
            pos = blob.find(MAGIC, pos + len(MAGIC))


def _notes(elf):
    for typ, off, size in sum((v for k, v in _sections(elf).items() if k.startswith(".note")), []):
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0").decode()
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernel_metadata(so_path):
    """{demangled-ish kernel symbol: metadata dict} for every kernel in the library"""
    data = open(so_path, "rb").read()
    out = {}
    for triple, elf in _code_objects(data):
        for name, ntype, desc in _notes(elf):
            if name == "AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
    return out


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(here, "learning3d_amd", "libl3d_hip.so")
    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    rows = kernel_metadata(so)
    print(f"{'kernel':70s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>8s} {'vspill':>7s}")
    for name in sorted(rows):
        if pat and not pat.search(name):
            continue
        k = rows[name]
        print(f"{name[:70]:70s} {k.get('.vgpr_count', 0):5d} {k.get('.sgpr_count', 0):5d} {k.get('.group_segment_fixed_size', 0):7d} "
              f"{k.get('.private_segment_fixed_size', 0):8d} {k.get('.vgpr_spill_count', 0):7d}")


if __name__ == "__main__":
    main()
