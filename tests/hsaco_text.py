"""Test infrastructure: what a reference code object (oracle/_ref/*.hsaco) CONTAINS, independent of where it was built.

`hipcc --genco` writes a clang offload bundle whose entries embed build paths, so the file's own hash differs between two builds
of the same source.  The machine code does not: text_sha256() unbundles the gfx950 entry (bundle header: magic, entry count, then
offset / size / triple per entry), finds the ELF64 section named .text and hashes it.  tests/golden/ref_hsaco_manifest.json holds
that hash per kernel file; tests/test_oracle.py rebuilds the code objects from /root/reference with the committed recipe and
compares (build container), tests/test_gpu_reference_kernels.py compares what it is about to load (GPU box)."""
import hashlib
import json
import os
import struct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANIFEST = os.path.join(ROOT, "tests", "golden", "ref_hsaco_manifest.json")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_elf(blob, arch="gfx950"):
    if blob[:4] == b"\x7fELF":
        return blob
    if blob[:len(MAGIC)] != MAGIC:
        raise ValueError("neither an ELF nor a clang offload bundle")
    (count,) = struct.unpack_from("<Q", blob, len(MAGIC))
    pos = len(MAGIC) + 8
    for _ in range(count):
        offset, size, id_len = struct.unpack_from("<QQQ", blob, pos)
        pos += 24
        triple = blob[pos:pos + id_len].decode()
        pos += id_len
        if triple.startswith("hip") and triple.endswith(arch) and size:
            return blob[offset:offset + size]
    raise ValueError(f"no {arch} entry in the bundle")


def elf_section(elf, name):
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        raise ValueError("not a little-endian ELF64")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    def header(i):
        return struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)  # name, type, flags, addr, offset, size, ...
    str_off, str_size = header(shstrndx)[4], header(shstrndx)[5]
    strtab = elf[str_off:str_off + str_size]
    for i in range(shnum):
        h = header(i)
        end = strtab.index(b"\0", h[0])
        if strtab[h[0]:end].decode() == name:
            return elf[h[4]:h[4] + h[5]]
    raise ValueError(f"no section {name}")


def text_sha256(path):
    with open(path, "rb") as f:
        return hashlib.sha256(elf_section(device_elf(f.read()), ".text")).hexdigest()


def manifest():
    with open(MANIFEST) as f:
        return json.load(f)


if __name__ == "__main__":   # python tests/hsaco_text.py > tests/golden/ref_hsaco_manifest.json  (in the build container, after make -C oracle)
    ref = os.path.join(ROOT, "oracle", "_ref")
    names = sorted(n for n in os.listdir(ref) if n.endswith(".hsaco"))
    print(json.dumps({"_what": "sha256 of the .text section of the gfx950 code object inside oracle/_ref/<file>: the reference's "
                               "src/kernels/<name>.cu compiled unmodified by oracle/Makefile (hipcc of ROCm 7.2.0); tests/hsaco_text.py",
                      **{n: text_sha256(os.path.join(ref, n)) for n in names}}, indent=1))
