"""Static checks on the gfx950 code objects inside the built library (no GPU needed).

A launch bound that is one step too tight makes hipcc spill registers to scratch without a word, and a staging array sized
for the worst case costs registers in every launch; both showed up only as lost step time.  These tests read the kernel
metadata (the msgpack note of each code object in the library's .hip_fatbin section) and pin what the step time depends on.
"""
import os
import struct

import msgpack
import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "semantic-segmentation_amd", "libssmi.so")


def _sections(buf):
    """name -> [(type, offset, size)] of a little-endian ELF64 image."""
    shoff, = struct.unpack_from("<Q", buf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", buf, 0x3A)

    def header(i):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", buf, shoff + i * shentsize)
        return name, typ, off, size
    _, _, stroff, _ = header(shstrndx)
    out = {}
    for i in range(shnum):
        name, typ, off, size = header(i)
        end = buf.index(b"\0", stroff + name)
        out.setdefault(buf[stroff + name:end].decode(), []).append((typ, off, size))
    return out


def _code_objects(path):
    """The gfx950 device images of every offload bundle linked into the library."""
    buf = open(path, "rb").read()
    (_, off, size), = _sections(buf)[".hip_fatbin"]
    fat = buf[off:off + size]
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = fat.find(magic)
    while pos >= 0:
        count, = struct.unpack_from("<Q", fat, pos + len(magic))
        p = pos + len(magic) + 8
        for _ in range(count):
            o, s, tl = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and s:
                yield fat[pos + o:pos + o + s]
        pos = fat.find(magic, pos + 1)


def _kernels(path):
    out = []
    for elf in _code_objects(path):
        for _typ, off, size in _sections(elf).get(".note", []):
            p = off
            while p < off + size:
                namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
                p += 12
                name = elf[p:p + namesz]
                p += (namesz + 3) & ~3
                desc = elf[p:p + descsz]
                p += (descsz + 3) & ~3
                if name.startswith(b"AMDGPU") and ntype == 32:      # NT_AMDGPU_METADATA
                    out.extend(msgpack.unpackb(desc, raw=False)["amdhsa.kernels"])
    return out


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libssmi.so is not built")
    ks = _kernels(LIB)
    assert len(ks) > 250, "expected the kernels of seven translation units, found {}".format(len(ks))
    return ks


def test_no_kernel_spills_registers_or_uses_scratch(kernels):
    bad = [(k[".name"], k[".vgpr_count"], k[".vgpr_spill_count"], k[".private_segment_fixed_size"]) for k in kernels
           if k[".vgpr_spill_count"] or k[".private_segment_fixed_size"]]
    assert not bad, "kernels with register spills / scratch: {}".format(bad[:5])


def test_every_kernel_is_wave64_with_at_most_256_threads(kernels):
    for k in kernels:
        assert k[".wavefront_size"] == 64, k[".name"]
        assert k[".max_flat_workgroup_size"] <= 1024, k[".name"]


def _named(kernels, prefix):
    return [k for k in kernels if k[".name"].startswith(prefix)]


def test_weight_gradient_slot_classes_keep_their_register_budget(kernels):
    """The strip kernels share the SIMDs with the main stream's kernels: the small slot classes must leave room for them
    (one tile per wave, 16-byte staging: at most 144 registers incl. accumulators, where the full class needs > 190)."""
    pre = "_ZN4ssmi19conv_bwd_weight_ldsI"
    small, full, prebn = [], [], []
    for k in _named(kernels, pre):
        args = k[".name"][len(pre):]
        # <T, TPW, VEC, M16, XV, YV, PRE> mangled as e.g. fLi1ELb1ELb0ELi8ELi2ELb0E
        if "Li1ELb1E" not in args:
            continue
        regs = k[".vgpr_count"] + k[".agpr_count"]
        if "ELi8ELi2ELb0EE" in args or "ELi2ELi8ELb0EE" in args:
            small.append(regs)
        elif "ELi12ELi8ELb0EE" in args:
            full.append(regs)
        elif "ELi2ELi8ELb1EE" in args:      # normalise-on-load variants (float32, the 1x1 expansions' class)
            prebn.append(regs)
    assert len(small) == 8 and len(full) == 4 and len(prebn) == 2, (small, full, prebn)      # 2 dtypes x M16 x 2 classes / 2 dtypes x M16
    assert max(small) <= 144 and max(prebn) <= 144, (small, prebn)
    assert min(full) > max(small), (small, full)


def test_direct_convolution_kernels_fit_their_launch_bounds(kernels):
    """conv_fwd_kernel<T, 1, VEC> is bound to four workgroups per CU: 128 registers (its merge-backward-on-load
    instantiation, <float, 1, true, 2>, to three: 168)."""
    ks = [k for k in _named(kernels, "_ZN4ssmi15conv_fwd_kernelI") if "Li1ELb" in k[".name"][:len("_ZN4ssmi15conv_fwd_kernelI") + 12]]
    plain = [k for k in ks if "Li1ELb1ELi2E" not in k[".name"] and "Li1ELb0ELi2E" not in k[".name"]]
    assert len(ks) == 5 and len(plain) == 4
    for k in ks:
        assert k[".vgpr_count"] + k[".agpr_count"] <= (128 if k in plain else 168), (k[".name"], k[".vgpr_count"])
