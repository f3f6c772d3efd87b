"""Register / scratch table of every gfx950 kernel of libpdgn_hip.so (no GPU needed).

    python tools/spill_table.py [--all] [--json out.json]            the code objects INSIDE the built library (seconds)
    python tools/spill_table.py --compile [--file gemm_x3.hip] ...     hipcc -Rpass-analysis=kernel-resource-usage (minutes)

Prints one line per kernel instance (demangled): VGPRs, AGPRs, SGPR / VGPR spill counts, scratch bytes per lane, LDS.
Default: only the instances that spill or use scratch; --all prints every one.  tests/test_spills.py runs the first
form and fails on an instance the default path can launch with vgpr_spill_count > 0."""
import argparse
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pdgn_amd", "csrc")
sys.path.insert(0, ROOT)

FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
          "LDS Size [bytes/block]": "lds"}


def _demangle(names):
    if not names:
        return {}
    filt = "c++filt"
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def scan(src):
    """[{name, vgprs, agprs, sgpr_spill, vgpr_spill, scratch, occupancy, lds, file}] of one .hip source."""
    from pdgn_amd import build as hip_build
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + hip_build.FLAGS + hip_build.EXTRA_FLAGS.get(os.path.basename(src), []) + \
        ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1), "file": os.path.basename(src)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+) \[-Rpass", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            v = m.group(2)
            cur[FIELDS[m.group(1).strip()]] = int(v) if v.isdigit() else v
    names = _demangle([r["mangled"] for r in rows])
    for r in rows:
        r["name"] = names.get(r["mangled"], r["mangled"])
    return rows


def scan_all(files=None, jobs=4):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and (not files or f in files))
    with ThreadPoolExecutor(jobs) as ex:
        return [r for rows in ex.map(scan, srcs) for r in rows]


def scan_built(so=None):
    """The same table from the AMDGPU metadata notes of the code objects bundled in the built library: what actually ships."""
    import tempfile
    from pdgn_amd import build as hip_build
    so = so or hip_build.build()
    llvm = "/opt/rocm/lib/llvm/bin/"
    rows = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([llvm + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        data = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        pos = [m.start() for m in re.finditer(re.escape(magic), data)] + [len(data)]
        for i in range(len(pos) - 1):
            part, co = os.path.join(d, "b%d.bin" % i), os.path.join(d, "co%d.o" % i)
            open(part, "wb").write(data[pos[i]:pos[i + 1]])
            r = subprocess.run([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + part, "--output=" + co], capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(r.stderr)
            notes = subprocess.run([llvm + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in re.split(r"\n  - \.agpr_count:", "\n" + notes)[1:]:
                blk = ".agpr_count:" + blk
                get = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
                rows.append({"mangled": re.search(r"\.name:\s+(\S+)", blk).group(1), "vgprs": get("vgpr_count"), "agprs": get("agpr_count"),
                             "sgprs": get("sgpr_count"), "sgpr_spill": get("sgpr_spill_count"), "vgpr_spill": get("vgpr_spill_count"),
                             "scratch": get("private_segment_fixed_size"), "lds": get("group_segment_fixed_size"), "file": "bundle %d" % i})
    names = _demangle([r["mangled"] for r in rows])
    for r in rows:
        r["name"] = names.get(r["mangled"], r["mangled"])
    return rows


def short(name, n=110):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", action="append")
    ap.add_argument("--compile", action="store_true")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--json")
    a = ap.parse_args()
    rows = scan_all(a.file) if a.compile or a.file else scan_built()
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
    print("%-112s %5s %5s %6s %6s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "Sspill", "Vspill", "scratchB", "occ", "LDS"))
    bad = 0
    for r in sorted(rows, key=lambda r: (-r.get("vgpr_spill", 0), -r.get("scratch", 0), r["name"])):
        dirty = r.get("vgpr_spill", 0) > 0 or r.get("scratch", 0) > 0
        bad += r.get("vgpr_spill", 0) > 0
        if a.all or dirty:
            print("%-112s %5s %5s %6s %6s %8s %4s %7s" % (short(r["name"]), r.get("vgprs"), r.get("agprs"), r.get("sgpr_spill"),
                                                        r.get("vgpr_spill"), r.get("scratch"), r.get("occupancy"), r.get("lds")))
    print("%d kernels, %d with spilled vector registers" % (len(rows), bad))


if __name__ == "__main__":
    main()
