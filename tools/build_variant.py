#!/usr/bin/env python3
"""Build an alternative libspatialclip_hip_<name>.so in which ONE source is compiled with extra -D flags (A/B of a kernel
variant on one GPU box: SC_HIP_LIB=<path> selects the library at import time; the other objects are the default build's).

    python tools/build_variant.py <name> <source.hip> -DFLAG [-DFLAG2 ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spatial-clip_amd")


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    sys.path.insert(0, PKG)
    import build as B
    B.build(verbose=False)
    objs = [os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B._sources() if s != src]
    vobj = os.path.join(B.OBJ, f"{src[:-4]}__{name}.o")
    subprocess.run([B.HIPCC, *B.FLAGS, *flags, "-c", os.path.join(B.CSRC, src), "-o", vobj], check=True)
    out = os.path.join(PKG, "lib", f"libspatialclip_hip_{name}.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, vobj, "-o", out], check=True)
    print(out)


if __name__ == "__main__":
    main()
