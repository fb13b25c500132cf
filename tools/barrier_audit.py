"""Static check of the generated gfx950 code: no LDS access may still be in flight (issued, its lgkmcnt wait not yet executed) when a
wave reaches an s_barrier.  `__builtin_amdgcn_s_barrier()` is not a memory fence for the compiler: in straight-line code it sinks the wait
for the last ds_read of a ring stage below the barrier, and the refill a faster wave issues right behind the barrier targets exactly that
buffer -- attn_qkv_fwd_bf16_kernel did this until round 4 (5 barriers per kernel; wrong output for a whole (cloud, head) in ~0.2 % of
launches under a concurrent load: tools/kernel_stress.py).  Every csrc/*.hip is compiled to assembly (hipcc -S, device only) and scanned
basic block by basic block.
    python tools/barrier_audit.py [file.hip ...]      exit code 1 and one line per finding; prints "clean" otherwise"""
import concurrent.futures, glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gm3d_amd", "csrc")


def assembly(src, outdir):
    out = os.path.join(outdir, os.path.basename(src)[:-4] + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                           "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    return out


def findings(path):
    """[(kernel, line, LDS accesses in flight)] : per basic block, ds_read / ds_write issued since the last lgkmcnt wait at an s_barrier"""
    kern, out, found = None, 0, []
    for i, l in enumerate(open(path).read().splitlines()):
        m = re.match(r"^(_Z\w+|gm3d\w+):", l)
        if m:
            kern, out = m.group(1), 0
            continue
        t = l.strip()
        if re.match(r"^\.LBB", t) or t.startswith(("s_cbranch", "s_branch")):
            out = 0
        elif t.startswith(("ds_read", "ds_write")):
            out += 1
        elif t.startswith("s_waitcnt"):
            mm = re.search(r"lgkmcnt\((\d+)\)", t)
            if mm:
                out = min(out, int(mm.group(1)))
            elif "vmcnt" not in t and "expcnt" not in t:
                out = 0
        elif t.startswith("s_barrier") and out > 0 and kern:
            found.append((kern, i + 1, out))
    return found


def audit(sources=None, jobs=4):
    sources = sources or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with tempfile.TemporaryDirectory(prefix="barrier_audit_") as d:
        with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
            paths = list(ex.map(lambda s: assembly(s, d), sources))
        return [(os.path.basename(p),) + f for p in paths for f in findings(p)]


if __name__ == "__main__":
    bad = audit([os.path.abspath(a) for a in sys.argv[1:]] or None)
    for b in bad:
        print("%s: %s line %d: %d LDS access(es) in flight at s_barrier" % b)
    print("clean" if not bad else "%d finding(s)" % len(bad))
    sys.exit(1 if bad else 0)
