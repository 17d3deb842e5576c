#!/usr/bin/env python3
"""How much of nvBowtie builds "unchanged" against the drop-in layer (north star: "so nvBowtie and sw-benchmark link unchanged").

Build container only (needs /root/reference).  Compiles every translation unit under nvBowtie/ AS IT LIES with
    hipcc --offload-arch=gfx950 -std=c++17 -fopenmp -include tools/port_cuda_calls.h -I include/nvbio_hip/compat -I <dir exposing only nvBowtie/>
so that every <nvbio/...> include resolves to the drop-in layer (never to the reference's library headers) and every <nvBowtie/...> include to the
reference's application sources, read in place.  Writes per-TU verdicts and the first errors of each failing TU to profiles/r04/nvbowtie_tu_check.log.

    python tools/nvbowtie_tu_check.py [--only reduce.cu select.cu ...] [--jobs 8] [--errors 6]
"""
import argparse
import concurrent.futures
import glob
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--errors", type=int, default=6)
    ap.add_argument("--log", default=os.path.join(ROOT, "profiles", "r04", "nvbowtie_tu_check.log"))
    ap.add_argument("--link", action="store_true", help="also link the objects (+ contrib/crc/crc.cpp) against libnvbio_hip.so into oracle/_ref/ref_nvBowtie")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("nvbowtie_tu_check: %s is not here (build container only)" % REF)
        return 0
    # the translation units of the reference's own build (the CMakeLists of nvBowtie/, bowtie2/ and bowtie2/cuda/): files that merely lie in
    # the tree but are in no target (score.cu includes a header that does not exist) are not part of nvBowtie
    tus = []
    for cm in sorted(glob.glob(os.path.join(REF, "nvBowtie", "**", "CMakeLists.txt"), recursive=True)):
        for line in open(cm, errors="replace"):
            name = line.strip()
            if (name.endswith(".cu") or name.endswith(".cpp")) and os.path.exists(os.path.join(os.path.dirname(cm), name)):
                tus.append(os.path.join(os.path.dirname(cm), name))
    tus = sorted(set(tus))
    if args.only:
        tus = [t for t in tus if os.path.basename(t) in args.only]
    tmp = tempfile.mkdtemp(prefix="nvbtu_")
    os.symlink(os.path.join(REF, "nvBowtie"), os.path.join(tmp, "nvBowtie"))
    os.symlink(os.path.join(REF, "contrib", "crc"), os.path.join(tmp, "crc"))                  # third-party CRC the application includes as <crc/crc.h>

    def build(tu):
        t0 = time.time()
        obj = os.path.join(tmp, os.path.basename(tu) + ".o")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O2" if args.link else "-O1", "-fopenmp", "-w", "-include", os.path.join(ROOT, "tools", "port_cuda_calls.h"),
                            "-x", "hip", "-c", tu] + (["-DNVBIO_HIP_COMPAT_DEBUG_TEXT"] if os.environ.get("NVBOWTIE_DEBUG_BUILD") and (os.path.basename(tu) == "reduce.cu" or os.path.basename(tu).startswith("aligner_")) else ["-DNVBIO_HIP_COMPAT_DEBUG_FULL"] if os.environ.get("NVBOWTIE_DEBUG_BUILD") else []) + os.environ.get("NVBOWTIE_EXTRA_FLAGS", "").split() + [ "-I" + os.path.join(ROOT, "include", "nvbio_hip", "compat"), "-I" + tmp, "-I" + os.path.join(ROOT, "tools", "port"), "-o", obj], capture_output=True, text=True)
        errs = [l.replace(tmp + "/", "").replace(ROOT + "/", "") for l in r.stderr.splitlines() if "error:" in l or "fatal error" in l]
        return tu, r.returncode == 0, errs, time.time() - t0

    out = ["nvbowtie_tu_check  %s" % time.strftime("%Y-%m-%d %H:%M:%S"),
           "every nvBowtie translation unit compiled as it lies: hipcc --offload-arch=gfx950 -std=c++17 -fopenmp -include tools/port_cuda_calls.h -I include/nvbio_hip/compat -I <nvBowtie/ and contrib/crc/ only> -I tools/port", ""]
    ok_n = 0
    with concurrent.futures.ThreadPoolExecutor(args.jobs) as ex:
        for tu, ok, errs, dt in ex.map(build, tus):
            ok_n += ok
            out.append("[%s] %s  (%.0f s)" % ("PASS" if ok else "FAIL", os.path.relpath(tu, REF), dt))
            seen = []
            for e in errs:
                if e not in seen:
                    seen.append(e)
            out += ["       " + e for e in seen[:args.errors]]
    out += ["", "%d / %d translation units compile unchanged" % (ok_n, len(tus))]
    if args.link and ok_n == len(tus) and not args.only:
        exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        crc = os.path.join(tmp, "crc.o")
        r = subprocess.run(["g++", "-O2", "-w", "-c", os.path.join(REF, "contrib", "crc", "crc.cpp"), "-I" + os.path.join(REF, "contrib"), "-o", crc], capture_output=True, text=True)
        objs = [os.path.join(tmp, os.path.basename(t) + ".o") for t in tus] + [crc]
        t0 = time.time()
        if r.returncode == 0:
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-fopenmp"] + objs + ["-L" + os.path.join(ROOT, "nvbio_amd", "lib"), "-lnvbio_hip", "-lz", "-lpthread",
                                "-Wl,-rpath,$ORIGIN/../../nvbio_amd/lib", "-o", exe], capture_output=True, text=True)
        errs = [l.replace(tmp + "/", "") for l in r.stderr.splitlines() if "error" in l or "undefined" in l][:20]
        out += ["", "[%s] link: %d objects + contrib/crc/crc.cpp + libnvbio_hip.so -> oracle/_ref/ref_nvBowtie  (%.0f s)" % ("PASS" if r.returncode == 0 else "FAIL", len(objs) - 1, time.time() - t0)]
        out += ["       " + e for e in errs]
    text = "\n".join(out) + "\n"
    if not args.only:
        os.makedirs(os.path.dirname(args.log), exist_ok=True)
        open(args.log, "w").write(text)
    failed = ok_n != len(tus) or (args.link and not args.only and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")))
    print(text)
    subprocess.run(["rm", "-rf", tmp])
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
