#!/usr/bin/env python3
"""In-tree build driver (no cmake needed): `python sibeliaz_amd/build.py [tools|lib|cli|oracle|all]`.

  tools   g++    -> sibeliaz_amd/bin/lcb-synth, lcb-mkgraph      (input generators)
  lib     hipcc  -> sibeliaz_amd/libsibeliaz_amd.so              (HIP kernels + C-ABI, gfx950)
  cli     hipcc  -> sibeliaz_amd/bin/sibeliaz-lcb                (drop-in executable)
  oracle  gcc    -> oracle/liblcb_oracle.so, oracle/lcb_oracle   (test infrastructure)
          + oracle/_ref when /root/reference exists (build container only)

Everything is rebuilt only when a source is newer than its output. hipcc cross-compiles gfx950
without a GPU, so this runs in the build container; the built files travel to the GPU box.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
BIN = os.path.join(PKG, "bin")
LIB = os.path.join(PKG, "libsibeliaz_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

LIB_SRC = ["graph.cpp", "bundles.cpp", "commit.cpp", "engine.cpp", "output.cpp", "capi.cpp", "comm.hip", "device.hip"]
LIB_HDR = ["lcb_host.h", "lcb_kernel.h", "lcb_device.h", "lcb_segments.h", "lcb_kernel_limits.h"]


def _newer(srcs, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_tools():
    os.makedirs(BIN, exist_ok=True)
    for name, src in (("lcb-synth", "synth.cpp"), ("lcb-mkgraph", "mkgraph.cpp")):
        s = os.path.join(CSRC, "tools", src)
        o = os.path.join(BIN, name)
        if _newer([s], o):
            _run(["g++", "-O2", "-std=c++17", "-fopenmp", "-o", o, s])


def hip_flags():
    # -disable-promote-alloca-to-lds: the compiler otherwise moves a 48-byte per-lane stack object of the process kernels into
    # LDS (48 B x 1024 lanes = 48 KB of the wide variant's budget); LDS is what bounds the seeds in flight per CU
    # -simplifycfg-sink-common=false: the compiler otherwise merges the stores of early exits (`S.status = ...; return`) with the last
    # store of the normal path (another field of the per-path state) into ONE store through a pointer that is either field - which
    # keeps both fields in scratch memory for the whole kernel (every `if (S.status)` a scratch load). Without it the shipped
    # instantiations use no scratch memory at all and 107 instead of 143 VGPRs (compact), no VGPR spills (wide, big).
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-Wall", "-Wno-unused-result", "-mllvm", "-disable-promote-alloca-to-lds",
            "-mllvm", "-simplifycfg-sink-common=false", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def build_lib(out=LIB, defines=()):
    """defines: extra -D macros (experiment variants: python sibeliaz_amd/build.py variant <name> <-D...>)."""
    srcs = [os.path.join(CSRC, s) for s in LIB_SRC]
    deps = srcs + [os.path.join(CSRC, h) for h in LIB_HDR] + [os.path.join(ROOT, "include", "lcb.h")]
    if _newer(deps, out):
        cmd = [HIPCC] + hip_flags() + list(defines) + ["-shared", "-o", out, "-ldl"]
        for s in srcs:
            if s.endswith(".hip"):
                cmd += ["-x", "hip", s]
            else:
                cmd += ["-x", "c++", s]
        _run(cmd)


def build_cli():
    build_lib()
    os.makedirs(BIN, exist_ok=True)
    s = os.path.join(CSRC, "main.cpp")
    o = os.path.join(BIN, "sibeliaz-lcb")
    if _newer([s, LIB], o):
        _run([HIPCC] + hip_flags() + ["-x", "c++", s, "-o", o, "-L" + PKG, "-lsibeliaz_amd", "-Wl,-rpath,$ORIGIN/.."])


def build_oracle():
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if os.path.isdir("/root/reference/SibeliaZ-LCB"):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])


def main(argv):
    if argv and argv[0] == "variant":      # experiment builds: libsibeliaz_amd_<name>.so with extra macros; use with LCB_LIB=<path>
        build_lib(os.path.join(PKG, "libsibeliaz_amd_%s.so" % argv[1]), argv[2:])
        return
    targets = argv or ["all"]
    for t in targets:
        if t == "tools":
            build_tools()
        elif t == "lib":
            build_lib()
        elif t == "cli":
            build_cli()
        elif t == "oracle":
            build_oracle()
        elif t == "all":
            build_tools()
            build_lib()
            build_cli()
            build_oracle()
        else:
            raise SystemExit("unknown target " + t)


if __name__ == "__main__":
    main(sys.argv[1:])
