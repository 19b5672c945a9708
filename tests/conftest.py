import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_addoption(parser):
    parser.addoption("--parity-report-only", action="store_true", default=False,
                     help="diagnostic run: the float64-yardstick asserts of tests/parity_tol.py become log lines; the parity report is stamped "
                          '"authoritative": false (tools/parity_stats.py reads such reports; the driver never passes this option)')


def pytest_sessionstart(session):
    """Safety net: if the in-tree library is missing or older than its sources, (re)build it before any test loads it (hipcc cross-compiles
    gfx950 without a GPU; unchanged translation units are skipped)."""
    from mixermdm_amd import build as B
    if B.needs_build():
        B.build(verbose=False)
    if session.config.getoption("--parity-report-only"):
        import parity_tol
        parity_tol.REPORT_ONLY = True


def pytest_sessionfinish(session, exitstatus):
    """Parity report of a GPU run (tests/parity_tol.py::REPORT): every compared step's out-of-tolerance fraction, masked persons, largest
    conditioning factor and the float64-yardstick quantiles.  gpurun merges gpurun_out/ back; the committed copy lives under profiles/."""
    try:
        import json
        import parity_tol
        if parity_tol.REPORT:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            from mixermdm_amd.build import sources_sha
            head = None
            try:                                    # the GPU box gets a snapshot without .git: the stamp file tools/_head.txt travels instead
                import subprocess
                head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or None
            except Exception:
                pass
            if not head and os.path.exists(os.path.join(ROOT, "tools", "_head.txt")):
                head = open(os.path.join(ROOT, "tools", "_head.txt")).read().strip()
            with open(os.path.join(ROOT, "gpurun_out", "parity_report.json"), "w") as f:
                # turned-joint events per precision mode: compare_step entries carry "[mode]" in their label
                events = {}
                for e in parity_tol.REPORT:
                    if e.get("kind") == "step_vs_fp32_oracle":
                        m = e["what"].rsplit("[", 1)[-1].rstrip("]") if e["what"].endswith("]") else "fp32"
                        events[m] = events.get(m, 0) + int(e.get("turned_joint_events", 0) > 0)
                json.dump({"exitstatus": int(exitstatus), "authoritative": not parity_tol.REPORT_ONLY, "git_head": head,
                           "kernel_sources_sha": {m: sources_sha(m) for m in ("fp32", "fp32_split", "bf16_fp8")},
                           "tests_selected": len(session.items), "events_by_mode": events, "entries": parity_tol.REPORT}, f, indent=1)
    except Exception as e:      # the report must never turn a green run red
        print("parity report not written:", e)


def load_golden(name):
    """Return (arrays, weights-by-prefix-getter) for tests/golden/<name>.npz."""
    import torch
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs = {k: d[k] for k in d.files}

    def weights(prefix, dims=()):
        """state_dict under 'w:<prefix>' as torch tensors, with the pe buffers re-created for `dims`."""
        from oracle.layers import pe_table
        W = {k[len("w:" + prefix):]: torch.from_numpy(v) for k, v in arrs.items() if k.startswith("w:" + prefix)}
        for key, D in dims:
            W[key] = pe_table(D)
        return W

    def t(key):
        return torch.from_numpy(np.asarray(arrs[key]))

    return arrs, weights, t


@pytest.fixture(scope="session")
def golden():
    return load_golden


def fulldims_case():
    """tests/golden/fulldims.npz: the reference at the real model sizes.  The fixture carries seeds, statistics and reference outputs only;
    weights and inputs are re-drawn here from the same seeded CPU generators make_golden.py used.  Returns (g, W, stats, inputs)."""
    import torch
    from oracle.layers import pe_table
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_inputs, FULL_DIMS
    d = np.load(os.path.join(GOLDEN, "fulldims.npz"))
    g = {k: d[k] for k in d.files}
    sd = synthetic_state_dict(seed=int(g["weights_seed"]), std=float(g["weights_std"]), bias_std=float(g["weights_bias_std"]), **FULL_DIMS)
    W = dict(sd)
    W["sequence_pos_encoder.pe"] = pe_table(512)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    W["denoiser2.sequence_pos_encoder.pe"] = pe_table(1024)
    stats = tuple(torch.from_numpy(g[k]) for k in ["mean_hml", "std_hml", "mean_ih", "std_ih"])
    rnd = lambda seed, *shape: torch.randn(*shape, generator=torch.Generator().manual_seed(int(seed)))
    n, T = [int(v) for v in g["fwd_shape"]]
    s1, s2, s3 = [int(v) for v in g["fwd_seeds"]]
    cond = rnd(s3, n, 8 * 768)
    cond[n // 2:] = 0
    B, Ts = int(g["step_B"]), int(g["step_T"])
    cb, xT = synthetic_inputs(B, Ts)
    c300, x300 = synthetic_inputs(1, 300)
    la, lb = [int(v) for v in g["late_seeds"]]
    inputs = dict(fwd=(rnd(s1, n, T, 524), rnd(s2, n, T, 524), cond, int(g["fwd_t"])),
                  step=(cb, xT, rnd(int(g["step_x2_seed"]), B, Ts, 524)),
                  t300=(c300, x300), late=(rnd(la, 1, 300, 524), rnd(lb, 1, 300, 524)))
    return g, sd, W, stats, inputs
