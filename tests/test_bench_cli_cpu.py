"""CPU: bench.py and __graft_entry__.py as the driver meets them — importable without a GPU, the contract's flags with the contract's
defaults (no flags = N = 1 and a K / W that finish within minutes), the N > 1 launcher's command line.  (The run itself needs a GPU.)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(argv):
    bench = importlib.import_module("bench")
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse_args()
    finally:
        sys.argv = old


def test_defaults_and_the_drivers_flags():
    a = _args([])
    assert a.gpus == 1 and a.steps == 2048 and a.warmup == 256 and a.envs == 4096
    assert a.backend == "nccl" and a.launch_form == "many" and a.steps_per_launch == 256 and a.pf_tol == 1e-12
    assert 0 < a.train_deadline < 180                    # below the process group's collective timeout (bench.py: 180 s)
    d = _args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (d.gpus, d.steps, d.warmup) == (8, 20, 5)


def test_the_line_names_baselines_metric_and_byte_convention():
    import json
    bench = importlib.import_module("bench")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert json.dumps(base["metric"])[1:-1] in src       # the metric string verbatim
    assert bench.B_ALG_CORE == 1340 and bench.B_ALG_WITH_OBS == 4220 and bench.HBM_PEAK_GBS == 8000.0
    for key in ('"roofline"', '"cpu_baseline"', '"value_device_events"', '"n_gpus"', '"ms_per_step"', '"scaling": "weak"',
                '"higher_is_better": True', '"dtype": "f64"', '"vs_baseline": None'):
        assert key in src, key


def test_graft_entry_has_build_and_smoke():
    ge = importlib.import_module("__graft_entry__")
    assert callable(ge.build) and callable(ge.smoke)
