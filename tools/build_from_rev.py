#!/usr/bin/env python3
"""Build libspatialclip_hip_<name>.so in which the named sources come from a git revision and every other object is the
working tree's (same-box A/B of the working tree against a committed state: SC_HIP_LIB=<path> selects the library).

    python tools/build_from_rev.py <name> <rev> <source.hip> [<source.hip> ...]
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spatial-clip_amd")


def main():
    name, rev, srcs = sys.argv[1], sys.argv[2], sys.argv[3:]
    sys.path.insert(0, PKG)
    import build as B
    B.build(verbose=False)
    objs = [os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B._sources() if s not in srcs]
    with tempfile.TemporaryDirectory() as tmp:
        for s in srcs:
            text = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:spatial-clip_amd/csrc/{s}"], check=True,
                                  capture_output=True, text=True).stdout
            path = os.path.join(tmp, s)
            with open(path, "w") as f:
                f.write(text)
            obj = os.path.join(B.OBJ, f"{s[:-4]}__{name}.o")
            subprocess.run([B.HIPCC, *B.FLAGS, "-I", B.CSRC, "-c", path, "-o", obj], check=True)   # headers: the working tree's
            objs.append(obj)
    out = os.path.join(PKG, "lib", f"libspatialclip_hip_{name}.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True)
    print(out)


if __name__ == "__main__":
    main()
