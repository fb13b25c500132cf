"""bench.py with module attributes overridden (A/B of the former environment knobs):
    python tools/bench_with.py fused.NOGRAD_SPLIT=1 gemm.FUSE_LN=True -- --steps 40 --warmup 10 --no-cpu-baseline"""
import importlib, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
for item in args[:cut]:
    name, val = item.split("=")
    mod, attr = name.rsplit(".", 1)
    if mod == "capi":          # capi.gm3d_ln_set_grid_cap=1024 -> call a process-wide knob of the C ABI
        import torch
        torch.cuda.init()           # (the HIP runtime first, as in bench.py: the library's first launch otherwise reports a stale error)
        torch.zeros(1, device="cuda")
        from gm3d_amd._capi import lib
        assert getattr(lib, attr)(int(eval(val))) == 0
        continue
    parts, obj = mod.split("."), None          # engine_pretrain.SegmentedDDPStep._reduce=... : module first, then attributes of it
    for cutp in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module("gm3d_amd." + ".".join(parts[:cutp]))
        except ImportError:
            continue
        for a in parts[cutp:]:
            obj = getattr(obj, a)
        break
    setattr(obj, attr, eval(val))
sys.argv = [os.path.join(ROOT, "bench.py")] + args[cut + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
