"""Identity of the kernel build a measurement belongs to: sha256 over ria_amd/csrc/* (names and contents, sorted) and the
compiler flags of ria_amd/build.py.  bench.py only quotes PMC-derived figures from a profiles/ file whose recorded hash equals
the current one."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_sha256():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, name)
        if os.path.isfile(p) and name.endswith((".hip", ".h", ".hpp", ".inc")):
            h.update(name.encode() + b"\0")
            with open(p, "rb") as f:
                h.update(f.read())
            h.update(b"\0")
    from .build import flags_line
    h.update(b"flags\0" + flags_line().encode())
    return h.hexdigest()
