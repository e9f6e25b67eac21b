#!/usr/bin/env python3
"""Headline benchmark: differentiable ICP iterations (fwd+bwd) on B=256 x 16384-point clouds.

    python bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2] ("C3"): per GPU B=256 synthetic scan pairs, n=m=16384,
point-to-plane + Huber(1.0) + soft trim(5.0), differentiable, dim=3, float32, T_init=I.
One "step" = one ICP iteration over the whole batch, forward AND backward: a timed call is
ONE differentiable icp() call of K constant iterations, backward of T.sum() w.r.t. source and
target, and (N>1) the RCCL all-gather of the poses, between barrier + device syncs.  Inputs are
resident in HBM when timing starts.  `value` is the MEDIAN of --reps (>= 5) such calls made
back to back after a steady-state warm-up ("warm-up 2 calls, then median of >= 5").

Beside `value` the same run reports, each as the median of its own timed calls:
  value_k10         the same call at K = 10 (SURVEY.md 8d's K) whenever --steps is something else: the cost of an
                    iteration depends on how far the clouds are from their pose, so numbers at different K do not compare
  value_bruteforce  the same K-iteration call with the brute-force kNN kernel in the loop (all n*m
                    pairs: the data-independent floor; `value` runs the exact slab-pruned search,
                    which returns the same indices but whose cost depends on the data and the pose)
  value_tolerance   a tolerance-mode call (tolerance 1e-4, up to 50 iterations, const_iter off):
                    cloud-iterations actually executed per second, early iterations included
  value_structured  the same K-iteration call on LiDAR-like scenes (ground plane + four walls, two of them
                    perpendicular to the sweep's sort axis, + 10 % clutter: dicp_amd/synthetic.make_scene_pairs)
  value_c2          BASELINE configs[1] (one GPU only): B=32 x 4096 points, point-to-point, K=10 fwd+bwd -- the Gauss-Newton loop
                    (`gn`) and the closed-form SVD step (`svd`: ICP.pt2pt_dICP_SVD), cloud-iterations/s and ms per call
  value_c4          BASELINE configs[3] on a 64-cloud slice (one GPU only): 65536-point clouds, pt2pl + Huber, K=5 fwd+bwd, with the exact
                    sweep (its plain searches score on the matrix cores from 32768 targets on) and with the matrix-core brute force in the loop
  ms_by_iteration_class   search + accumulate time of the event-carrying call by kind of iteration: full search (before the
                    certificates start), certifying search, certified (guard launch + accumulate with its on-the-spot searches)
`value_bruteforce` runs the matrix-core brute force (split-f16 filter on v_mfma_f32_32x32x16_f16 + exact float32 refine: the same indices);
`value_bruteforce_valu` is the same call with the float32 FMA kernel of rounds 1-3.

For N>1 launch with torch.distributed.run (one rank per GPU, weak scaling: 256 clouds per rank).
Rank 0 prints ONE JSON line.  Kernel times come from HIP events carried on the dispatches of the last
timed call; `roofline` is the kernel with the LARGEST SHARE of that call (shares are in the line), the
other two legs ride beside it: the search (`bound: "valu"`: f32 FMA work, 8 flop per scored pair against
the 157.3 TF f32 peak) and the HBM-bound forward / backward accumulate kernels on their ALGORITHMIC bytes
(48n / 88n per cloud; 44n / 84n when the weights are the implicit ones and nothing is read
for them).  `cpu_baseline` is the CPU oracle (the reference's op sequence) timed on this host for a
bounded sample.  The gate compares the TIMED call's own poses and gradients (K iterations, certificates
engaged) with the oracle run for the same K iterations on the first two clouds; the run FAILS (exit 1 on
every rank, value null) if a result of any rank is not finite or the pose differs by > 1e-4.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dicp_amd import dist as ddist                      # noqa: E402
from dicp_amd.ICP import ICP                            # noqa: E402
from dicp_amd.synthetic import make_pairs, make_scene_pairs, make_independent_pairs   # noqa: E402

STEADY_CALLS, STEADY_MAX = 3, 40   # untimed K-iteration calls before the timed ones: at least / at most
F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak
F16_MFMA_PEAK_TFLOPS = 2500.0   # ... dense f16 / bf16 MFMA peak
HBM_PEAK_GBS = 8000.0        # HBM3E spec
LOSS = {"name": "huber", "metric": 1.0}
TRIM = 5.0
POSE_BAR = 1e-4              # north-star parity bar for poses
GRAD_BAR = 1e-3              # ... and for gradients (of their scale)
EV_PER_ITER = 6              # dicp_loop_buffers.events: kNN, accumulate (forward), accumulate_bwd -- a start/stop pair each


class EventLog:
    """HIP events around the kNN, accumulate (forward) and accumulate_bwd (backward) launch of every iteration.
    The loop runs inside libdicp_hip.so (dicp_icp_forward / _backward), so the library records them: it is handed
    the raw hipEvent_t handles of these torch events (same HIP runtime) -- on the stream the kernels run on.  The
    sweep, accumulate and windowed-backward launches carry their pair on the dispatch itself (hipExtLaunchKernel
    start / stop events: the kernel's own begin / end timestamps, no barrier packets in the timed queue); the
    brute-force and atomic forms are bracketed by hipEventRecord."""
    OFF = {"knn": 0, "accumulate": 2, "accumulate_bwd": 4}

    def __init__(self):
        self.K, self.ev, self.arr = 0, None, None

    def handles(self, K):
        import ctypes
        if self.K != K:
            self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(EV_PER_ITER * K)]
            for e in self.ev:
                e.record()                       # materialises the underlying hipEvent_t
            self.arr = (ctypes.c_void_p * (EV_PER_ITER * K))(*[e.cuda_event for e in self.ev])
            self.K = K
        return ctypes.cast(self.arr, ctypes.c_void_p)

    def all_ms(self, name):
        off = self.OFF[name]
        out = []
        for k in range(self.K):
            try:
                out.append(self.ev[EV_PER_ITER * k + off].elapsed_time(self.ev[EV_PER_ITER * k + off + 1]))
            except Exception:
                return []
        return out


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def kernel_sources_sha16():
    """A hash of the library's sources (csrc + the header): what a committed PMC profile is stamped with (scripts/pmc_summary.py), so that a traffic figure
    taken from it can say whether the kernels have changed since (the GPU box has no .git to ask)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "dicp_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "dicp_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_prefix, B, n):
    """HBM bytes per launch from the COMMITTED rocprofv3 PMC passes (profiles/*_pmc_hbm_traffic*.json, made by
    scripts/pmc_summary.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command).  PMC counters
    cannot be read from inside the process: this number is NOT measured by the run that prints it, and the line says so
    (file and the commit the profile was taken at).  None when no matching profile exists."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != {"B": B, "n": n}:
            continue
        hits = [k for name, k in d["kernels"].items() if name.startswith(kernel_prefix) and k.get("launches")]
        if hits:        # several launch configurations of one kernel (template arguments): launch-weighted mean
            tot = sum(k["launches"] for k in hits)
            stale = d.get("kernel_sources_sha16") != kernel_sources_sha16()
            src = "from committed profile %s@%s (not measured by this run)%s" % (os.path.basename(f), d.get("commit", "unknown"),
                  "; STALE: the kernel sources have changed since that profile was taken" if stale else "; kernel sources unchanged since")
            if kernel_prefix.startswith("accumulate_bwd") and all("hbm_bytes_full_launches" in k for k in hits):      # (launches with every cloud at work)
                tot = sum(k["full_launches"] for k in hits)
                return sum(k["hbm_bytes_full_launches"] * k["full_launches"] for k in hits) / tot, src
            return sum(k["hbm_bytes"] * k["launches"] for k in hits) / tot, src
    return None, None


def run_call(icp, src, tgt, T0, world):
    """world: number of ranks, or -1 to run the collective even with a single rank (smoke test of the N>1 path)."""
    s = src.detach().requires_grad_(True)
    t = tgt.detach().requires_grad_(True)
    out = icp.icp(s, t, T0, trim_dist=TRIM, loss_fn=LOSS, dim=3)
    # every rank holds the same number of clouds: the shard sizes are known, no size exchange (and no host sync) before the gather.  The
    # poses are final once the forward is queued: the all-gather goes out on the communicator's stream NOW and runs under the backward
    T_all, work = (ddist.gather_poses_async(out["T"].detach(), total=src.shape[0] * abs(world), force=True) if world != 1 else (out["T"].detach(), None))
    out["T"].sum().backward()
    if work is not None:
        work.wait()
    return out, T_all, s.grad, t.grad


_T_START = time.perf_counter()


def progress(msg):
    """One line per leg on stderr (stdout carries the JSON line only): a long run shows where it is -- and, should a leg fail, which one."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %6.1f s] %s\n" % (time.perf_counter() - _T_START, msg))
        sys.stderr.flush()


def cpu_baseline(n, m, budget_s=25.0):
    """The oracle (reference op sequence: cdist -> argmin -> gather -> ... -> linalg.inv -> matrix_exp)
    on this host's cores, fwd+bwd, on a bounded sample: chunks of 4 clouds x 3 iterations
    ((4,n,m) fp32 distances = 4 GiB per chunk at 16384^2; per-cloud cost is flat in B), about 10 s of CPU work."""
    from oracle import dicp_oracle as O
    # 16 threads is the fastest setting for this op sequence on the GPU box's 2 x EPYC 9575F host
    # (tests/tools/cpu_threads_probe.py -> profiles/r01_cpu_threads_probe.txt: 8/16/32/64/128/256 threads -> 3.9/5.5/5.0/3.4/2.1/0.1 cloud-it/s)
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    Bc, K = 4, 3
    src, tgt = make_pairs(Bc, n, m, seed=3, dtype=torch.float32)
    T0 = torch.eye(4).repeat(Bc, 1, 1)
    times = []
    t_start = time.time()
    while len(times) < 4 and (time.time() - t_start) < budget_s:
        s, t = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
        t0 = time.time()
        ref = O.icp_batched(s, t, T0, torch.ones(Bc, n), icp_type="pt2pl", differentiable=True, max_iterations=K,
                            tolerance=1e-12, trim_dist=TRIM, loss_fn=LOSS, dim=3, const_iter=True, tanh_steepness=5.0)
        ref["T"].sum().backward()
        times.append(time.time() - t0)
    best = median(times)
    return {"value": Bc * K / best, "unit": "cloud-iterations/s", "cores": cores, "kind": "port",
            "sample": "%d clouds x %d iterations fwd+bwd, n=m=%d, float32, median of %d runs (%.2f s each)"
                      % (Bc, K, n, len(times), best)}


def other_configs(make_icp, dev, sync):
    """BASELINE configs[1] and a slice of configs[3] on this GPU: median of 5 (3) eager calls each, forward + backward, steady state first."""
    from dicp_amd import _lib as L
    out = {}

    def time_calls(fn, count, warm=3):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(count):
            sync()
            t0 = time.perf_counter()
            fn()
            sync()
            ts.append(time.perf_counter() - t0)
        return median(ts)

    # configs[1]: B = 32 x 4096, point-to-point, K = 10
    Bc, nc, Kc = 32, 4096, 10
    s2, t2 = make_pairs(Bc, nc, nc, seed=2, dtype=torch.float32)
    s2, t2 = s2.to(dev), t2[:, :, :3].contiguous().to(dev)
    T2 = torch.eye(4, device=dev).repeat(Bc, 1, 1)
    gn = make_icp(icp_type="pt2pt", differentiable=True, max_iterations=Kc, tolerance=1e-12)
    gn.const_iter = True

    def gn_call():
        s, t = s2.detach().requires_grad_(True), t2.detach().requires_grad_(True)
        gn.icp(s, t, T2, trim_dist=TRIM, loss_fn=LOSS, dim=3)["T"].sum().backward()

    def svd_call():
        s, t = s2.detach().requires_grad_(True), t2.detach().requires_grad_(True)
        gn.pt2pt_dICP_SVD(s, t, T2, trim_dist=TRIM)[1].sum().backward()
    t_gn, t_svd = time_calls(gn_call, 5), time_calls(svd_call, 5)
    out["value_c2"] = {"workload": "BASELINE configs[1]: B=32 x 4096-pt clouds, point-to-point, K=10 fwd+bwd, eager calls",
                       "gn": {"cloud_iterations_per_s": Bc * Kc / t_gn, "ms_per_call": t_gn * 1e3},
                       "svd": {"cloud_iterations_per_s": Bc * Kc / t_svd, "ms_per_call": t_svd * 1e3, "note": "ICP.pt2pt_dICP_SVD: closed-form 3x3 SVD step (batched, weighted)"}}
    del s2, t2, gn
    # configs[3], a 64-cloud slice: 65536-point clouds, pt2pl + Huber, K = 5
    Bc, nc, Kc = 64, 65536, 5
    s4, t4 = make_pairs(Bc, nc, nc, seed=4, dtype=torch.float32)
    s4, t4 = s4.to(dev), t4.to(dev)
    T4 = torch.eye(4, device=dev).repeat(Bc, 1, 1)
    legs = {}
    for name, kv in (("sweep", L.KNN_AUTO), ("mfma_bruteforce", L.KNN_MFMA)):
        obj = make_icp(icp_type="pt2pl", differentiable=True, max_iterations=Kc, tolerance=1e-12)
        obj.const_iter, obj.knn_variant = True, kv

        def call():
            s, t = s4.detach().requires_grad_(True), t4.detach().requires_grad_(True)
            obj.icp(s, t, T4, trim_dist=TRIM, loss_fn=LOSS, dim=3)["T"].sum().backward()
        tc = time_calls(call, 3, warm=2)
        legs[name] = {"cloud_iterations_per_s": Bc * Kc / tc, "ms_per_call": tc * 1e3, "ms_per_iteration": tc * 1e3 / Kc}
        if name == "sweep" and "knn_pairs" in obj.knn_stats:
            legs[name]["pairs_scored_fraction"] = float(obj.knn_stats["knn_pairs"].sum().item()) / Kc / (float(Bc) * nc * nc)
        del obj
    del s4, t4, T4
    # configs[3] at its FULL batch: 256 x 65536, the default search (sorted sweep, matrix-core scoring from 32768 targets on)
    Bf = 256
    parts = [make_pairs(64, nc, nc, seed=4, dtype=torch.float32, first=64 * i) for i in range(Bf // 64)]
    sf, tf = torch.cat([p_[0] for p_ in parts]).to(dev), torch.cat([p_[1] for p_ in parts]).to(dev)
    del parts
    Tf = torch.eye(4, device=dev).repeat(Bf, 1, 1)
    full = make_icp(icp_type="pt2pl", differentiable=True, max_iterations=Kc, tolerance=1e-12)
    full.const_iter = True

    def full_call():
        s, t = sf.detach().requires_grad_(True), tf.detach().requires_grad_(True)
        o = full.icp(s, t, Tf, trim_dist=TRIM, loss_fn=LOSS, dim=3)
        o["T"].sum().backward()
        return o, s, t
    tfull = time_calls(full_call, 3, warm=2)
    o, s_, t_ = full_call()
    out["value_c4_full"] = {"workload": "BASELINE configs[3] at its full batch: B=256 x 65536-pt clouds, point-to-plane + huber(1.0) + trim(5.0), K=5 fwd+bwd, default search",
                            "cloud_iterations_per_s": Bf * Kc / tfull, "ms_per_call": tfull * 1e3, "ms_per_iteration": tfull * 1e3 / Kc,
                            "pairs_scored_fraction": float(full.knn_stats["knn_pairs"].sum().item()) / Kc / (float(Bf) * nc * nc) if "knn_pairs" in full.knn_stats else None,
                            "finite": bool(torch.isfinite(o["T"]).all() and torch.isfinite(s_.grad).all() and torch.isfinite(t_.grad).all()),
                            "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9}
    del sf, tf, Tf, full, o, s_, t_
    # configs[0]: the reference's own test pair (tests/test_ICP.py:35-117: 65 points, float64, point-to-plane, dim 2) -- latency of one call, forward + backward
    try:
        import numpy as np
        from dicp_amd.graphed import graphed_icp_step
        scan = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "points_scan.npy"))).to(torch.float64)
        mp = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "points_map.npy"))).to(torch.float64)
        s1, t1 = scan.to(dev), mp.to(dev)
        T1 = torch.eye(4, dtype=torch.float64, device=dev)
        one = make_icp(icp_type="pt2pl", differentiable=True, max_iterations=6, tolerance=1e-12)     # (6: where the reference's own tolerance stops this pair)
        one.const_iter = True
        kw1 = dict(trim_dist=5.0, loss_fn={"name": "huber", "metric": 10.0}, dim=2)

        def one_call():
            s, t = s1.detach().requires_grad_(True), t1.detach().requires_grad_(True)
            one.icp(s, t, T1, **kw1)["T"].sum().backward()
        t_eager = time_calls(one_call, 20, warm=5)
        sg, tg = s1.detach().unsqueeze(0).requires_grad_(True), t1.detach().unsqueeze(0).requires_grad_(True)
        step = graphed_icp_step(one, lambda o_: o_["T"].sum(), sg, tg, T1.unsqueeze(0), **kw1)
        t_graph = time_calls(lambda: step(sg, tg, T1.unsqueeze(0)), 20, warm=5)
        out["value_c1"] = {"workload": "BASELINE configs[0]: the reference's tests/data pair (%d source / %d target points), float64, point-to-plane + huber(10) + trim(5), dim 2, "
                                       "6 iterations, forward + backward of T.sum(), one pair per call" % (s1.shape[0], t1.shape[0]),
                           "eager_ms_per_call": t_eager * 1e3, "graphed_ms_per_call": t_graph * 1e3,
                           "note": "latency, not throughput: one block per cloud runs the whole loop (icp_small_* kernels); graphed = dicp_amd.graphed.graphed_icp_step (call + loss + backward as one hipGraph)"}
    except Exception as e:      # (the fixtures travel with tests/; a missing file must not cost the run its line)
        out["value_c1"] = {"error": repr(e)}
    legs["workload"] = "BASELINE configs[3] on a 64-cloud slice: 65536-pt clouds, point-to-plane + huber(1.0) + trim(5.0), K=5 fwd+bwd"
    legs["note"] = ("sweep: the exact sorted sweep, its plain searches scoring on the matrix cores (from 32768 targets per cloud on); mfma_bruteforce: all n*m pairs on "
                    "v_mfma_f32_32x32x16_f16 (split-f16 filter + exact float32 refine); matrix-pipe counters: profiles/r06_knn_c4_65536_pmc.txt")
    out["value_c4"] = legs
    return out


def oracle_gate(src, tgt, K, out_T, g_src, g_tgt):
    """The TIMED call's own results on its first clouds against the oracle run for the same K iterations (fwd+bwd):
    -> (dict for the line, ok).  Pose bar 1e-4; gradients: median row error <= 1e-5 and at most 0.1 % of the rows beyond 1e-3
    of the gradient's scale (a float32 argmin may pick the other of two nearly equidistant targets for a handful of queries)."""
    from oracle import dicp_oracle as O
    Bc, n = src.shape[0], src.shape[1]
    s, t = src.detach().cpu().clone().requires_grad_(True), tgt.detach().cpu().clone().requires_grad_(True)
    t0 = time.time()
    ref = O.icp_batched(s, t, torch.eye(4).repeat(Bc, 1, 1), torch.ones(Bc, n), icp_type="pt2pl", differentiable=True, max_iterations=K,
                        tolerance=1e-12, trim_dist=TRIM, loss_fn=LOSS, dim=3, const_iter=True, tanh_steepness=5.0)
    ref["T"].sum().backward()
    dT = float((out_T.detach().cpu() - ref["T"].detach()).abs().max())
    info = {"pose_max_abs_diff_vs_oracle": dT, "bar": POSE_BAR, "clouds": Bc, "iterations": K,
            "what": "the timed call's own T / source.grad / target.grad (clouds 0..%d) against the oracle run for the same %d iterations" % (Bc - 1, K),
            "oracle_seconds": None}
    ok = dT <= POSE_BAR
    for name, got, want in (("source", g_src, s.grad), ("target", g_tgt, t.grad)):
        scale = max(1.0, float(want.abs().max()))
        err = (got.detach().cpu() - want).abs().amax(dim=2)
        info["grad_%s_rows_beyond_1e-3" % name] = float((err > GRAD_BAR * scale).float().mean())
        info["grad_%s_median_row_err" % name] = float(err.median()) / scale
        ok = ok and info["grad_%s_rows_beyond_1e-3" % name] < 1e-3 and info["grad_%s_median_row_err" % name] < 1e-5
    info["oracle_seconds"] = round(time.time() - t0, 2)
    return info, ok


def main(argv=None, make_icp=None, device=None, backend="nccl", emit=None):
    """argv / make_icp / device / backend / emit: the launcher seam.  bench.py itself runs with the defaults (dicp_amd's ICP on
    this rank's MI355X, RCCL, the JSON line to stdout); tests/test_bench_launcher.py drives the very same launcher path -- rank
    environment, process group, weak-scaling seeds, barriers, max-over-ranks timing, pose all-gather, the JSON line -- on two gloo
    CPU ranks with a stand-in `make_icp`, which is how the N > 1 path is covered without a multi-GPU box."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="K: ICP iterations per timed call (SURVEY 8d: 10)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5, help="timed K-iteration calls; `value` is their median (>= 5)")
    ap.add_argument("--batch", type=int, default=256, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=16384, help="points per cloud (source and target)")
    ap.add_argument("--knn", choices=["auto", "sweep", "valu", "mfma"], default="auto",
                    help="auto/sweep: exact slab-pruned kNN (same indices as brute force); valu/mfma: brute-force kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip value_k10 / value_bruteforce / value_tolerance / value_structured")
    args = ap.parse_args(argv)
    head = git_head()                   # (a subprocess: before the GPU or the process group is touched)
    on_gpu = device is None
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    if make_icp is None:
        make_icp = lambda **kw: ICP(**kw)                               # noqa: E731

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
        if torch.cuda.device_count() < max(1, args.gpus) or local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py --gpus %d: this node shows %d device(s) (local rank %d)" % (args.gpus, torch.cuda.device_count(), local_rank))
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if on_gpu else device
    # synthetic inputs are generated on the CPU: keep N ranks from oversubscribing the host's cores
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 1) // max(1, world))))
    # DICP_BENCH_FORCE_DIST=1 runs the distributed code path (RCCL init, barrier, pose all-gather, max-reduce)
    # even with one rank: how the N>1 path is smoke-tested on a 1-GPU box
    force_dist = os.environ.get("DICP_BENCH_FORCE_DIST", "0") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    ranks_seen, rccl = 1, None
    if use_dist:
        if on_gpu:
            torch.distributed.init_process_group(backend, device_id=dev)        # "nccl" = RCCL
        else:
            torch.distributed.init_process_group(backend)
        ranks_seen = torch.distributed.get_world_size()
        if ranks_seen != max(1, args.gpus if args.gpus > 1 else world):
            raise SystemExit("bench.py --gpus %d: the process group has %d ranks" % (args.gpus, ranks_seen))
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        # one rank per device: two ranks on one GPU would time half a GPU each and report it as two
        mine = torch.tensor([local_rank if on_gpu else rank], dtype=torch.int64, device=dev if on_gpu else device)
        every = [torch.empty_like(mine) for _ in range(ranks_seen)]
        torch.distributed.all_gather(every, mine)
        seen_devices = sorted(int(v.item()) for v in every)
        if len(set(seen_devices)) != ranks_seen:
            raise SystemExit("bench.py --gpus %d: the ranks map to devices %s -- one rank per device is required" % (args.gpus, seen_devices))

    B, n, m, K, W = args.batch, args.points, args.points, args.steps, args.warmup
    reps = max(5, args.reps)
    src, tgt = make_pairs(B, n, m, seed=3, dtype=torch.float32, first=rank * B)
    src, tgt = src.to(dev), tgt.to(dev)
    T0 = torch.eye(4, device=dev).repeat(B, 1, 1)
    cw = world if not (force_dist and world == 1) else -1

    def fence():
        sync()
        if use_dist:
            torch.distributed.barrier()
        sync()

    def all_max(x):
        if not use_dist:
            return x
        tmax = torch.tensor([x], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.item())

    def all_ranks(x):
        """this rank's number on every rank, in rank order"""
        if not use_dist:
            return [x]
        mine = torch.tensor([x], dtype=torch.float64, device=dev)
        every = torch.empty((ranks_seen,), dtype=torch.float64, device=dev)
        torch.distributed.all_gather_into_tensor(every, mine)
        return [float(v) for v in every.tolist()]

    def steady(icp_obj, data, least=STEADY_CALLS, most=STEADY_MAX):
        """Untimed calls at the timed call's own shapes until three in a row agree to 3 %: the caching allocator then owns
        the K-sized buffers (the first use of a new size is a synchronous hipMalloc) and the chip is in its steady state
        (back-to-back calls get faster for a while on a cold box: scripts/call_repeat_diag.py).  Same count on every rank."""
        calls, recent, held = 0, [], None
        while calls < most:
            sync()
            t_w = time.perf_counter()
            held = run_call(icp_obj, data[0], data[1], T0, cw)      # (the previous call's results stay alive during the next one, as in timed(): the
            sync()                                                  #  allocator then owns room for two result sets before the first timed call)
            recent = (recent + [time.perf_counter() - t_w])[-3:]
            calls += 1
            done = calls >= least and max(recent) <= 1.03 * min(recent)
            if use_dist:
                flag = torch.tensor([1 if done else 0], device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                done = bool(flag.item())
            if done:
                break
        del held
        return calls

    def timed(icp_obj, data, count, events=None):
        """`count` calls, each bracketed by barrier + device sync on both sides; per call the MAX over ranks (and every rank's own time)."""
        times, by_rank, last = [], [], None
        for i in range(count):
            if events is not None:          # the kernel timings of the roofline legs: HIP events on the dispatches of the LAST timed call only
                icp_obj._tuning["timing_events"] = events if i == count - 1 else None
            fence()
            t0 = time.perf_counter()
            last = run_call(icp_obj, data[0], data[1], T0, cw)
            fence()
            mine = time.perf_counter() - t0
            times.append(all_max(mine))
            by_rank.append(all_ranks(mine))
        return times, last, by_rank

    def new_icp(K_, tol=1e-12, const_iter=True, knn=None):
        obj = make_icp(icp_type="pt2pl", differentiable=True, max_iterations=K_, tolerance=tol)
        obj.const_iter = const_iter
        obj.knn_variant = {"auto": 0, "sweep": 3, "valu": 1, "mfma": 2}[args.knn] if knn is None else knn
        return obj

    data = (src, tgt)
    icp = new_icp(max(W, 1))
    brute = args.knn in ("valu", "mfma")      # (auto also runs brute force when the clouds are small: settled after the timed calls)
    if W > 0:
        run_call(icp, src, tgt, T0, cw)                                  # W untimed warm-up steps
    icp.max_iterations = K
    log = EventLog()
    if on_gpu:
        log.handles(K)                      # create the HIP events now: not part of the timed workload
    # An event pair carried on a dispatch costs the queue ~4 us of idle time before the next dependent launch (6 % of a call at this shape:
    # profiles/r02_timed_call_timeline*.txt): the events ride on the last timed call only, `value` is the median of all of them.
    use_events = on_gpu and os.environ.get("DICP_BENCH_NO_EVENTS") != "1"
    icp._tuning["timing_events"] = None
    # a generational GC pass over this process's heap takes tens of ms (10 steps take 5 ms): whether one lands inside a
    # timed call depends on the allocation count so far, i.e. on things as irrelevant as argv -> collect now and pause
    # the collector.  The collection goes BEFORE the steady-state calls: the first call after one is ~0.8 ms slower
    # (it frees the previous calls' graphs and the caching allocator re-splits its blocks).
    gc.collect()
    gc.disable()
    steady_calls = steady(icp, data)
    times, (out, T_all, gs, gt), by_rank = timed(icp, data, reps, log if use_events else None)      # reps x exactly K steps
    elapsed = median(times)
    i_med = sorted(range(len(times)), key=lambda i: times[i])[len(times) // 2]
    ev_ms = {nm: (log.all_ms(nm) if on_gpu else []) for nm in ("knn", "accumulate", "accumulate_bwd")}       # of the LAST timed call
    sane = bool(torch.isfinite(out["T"]).all() and torch.isfinite(gs).all() and torch.isfinite(gt).all())
    from dicp_amd import _ops, _lib as L
    from dicp_amd._ops import auto_knn_kind
    # knn=auto takes the brute-force kernel for small clouds (no sorted-sweep statistics then)
    if (args.knn == "auto" and auto_knn_kind(B, n, m) != L.KNN_SWEEP) or not on_gpu:
        brute = True

    def sweep_stats(obj, K_):
        """pairs scored per kNN launch of the object's LAST call (each call has its own counters), and what its certified iterations searched again"""
        if brute or "knn_pairs" not in obj.knn_stats:
            return None, None
        pairs = float(obj.knn_stats["knn_pairs"].sum().item()) / K_
        again = obj.knn_stats.get("searched_again")     # (K,128) int32: units [0,64) / single queries [64,128) searched again
        cert = None
        if again is not None:
            cert = {"units_searched_again": [int(v) for v in again[:K_, :64].sum(1).tolist()], "queries_searched_again": [int(v) for v in again[:K_, 64:].sum(1).tolist()],
                    "units": B * ((n + 127) // 128), "queries": B * n,
                    "note": "match certificates: from the last re-ordering of the queries on (0, 0 = every query searched) a launch searches only the "
                            "units, and the accumulate only the single queries, whose match is not PROVEN unchanged; results identical to searching "
                            "everything (tests/test_gpu_configs.py::test_timed_loop_at_its_own_size)"}
        return pairs, cert

    pairs_scored, certified = sweep_stats(icp, K)
    # the backward's truncated reverse sweep: clouds at work per iteration of the last timed call (None: feature off / not this path)
    bwd_live = icp.knn_stats.get("bwd_live") if on_gpu else None
    bwd_live = [int(v) for v in bwd_live[:K].tolist()] if bwd_live is not None else None

    # ---- the other legs, in the same run (each the median of its own timed calls)
    progress("timed calls done (K = %d): %.3f ms per call" % (K, elapsed * 1e3))
    extra = {}
    if not args.no_extra_legs:
        if K != 10:
            progress("leg: value_k10")
            k10 = new_icp(10)
            steady(k10, data, least=3, most=8)
            kt, _, _ = timed(k10, data, reps)
            extra["value_k10"] = world * B * 10 / median(kt)
            extra["k10_note"] = "the same call at K = 10 (SURVEY.md 8d): %.3f ms per step, median of %d calls" % (median(kt) * 1e3 / 10, reps)
            del k10
        # A training loop presents NEW clouds of the same shape every step, and what an ICP object carries from call to call (CallHints: tail placement,
        # certificate pauses, the scoring forms' plan) was learned on other data.  The K = 10 call of ONE object rotating over four different batches
        # (every timed call's hints come from another batch); results do not depend on hints (tests/test_gpu_hints.py), time may.
        progress("leg: value_fresh_inputs")
        fresh = [data] + [tuple(x.to(dev) for x in make_pairs(B, n, m, seed=101 + i, dtype=torch.float32, first=rank * B)) for i in range(3)]
        fr = new_icp(10)
        for d_ in fresh + fresh:
            run_call(fr, d_[0], d_[1], T0, cw)
        ft = []
        for i in range(2 * len(fresh)):
            d_ = fresh[i % len(fresh)]
            fence()
            t0_ = time.perf_counter()
            run_call(fr, d_[0], d_[1], T0, cw)
            fence()
            ft.append(all_max(time.perf_counter() - t0_))
        extra["value_fresh_inputs"] = world * B * 10 / median(ft)
        extra["fresh_inputs_note"] = ("K = 10, one ICP object, %d timed calls rotating over %d different make_pairs batches of the benchmark's shape (seeds 3, 101-103): every "
                                      "call-to-call hint was recorded on another batch; %.3f ms per call (min %.3f, max %.3f); compare value_k10, whose calls replay one batch"
                                      % (len(ft), len(fresh), median(ft) * 1e3, min(ft) * 1e3, max(ft) * 1e3))
        del fr, fresh
        if world == 1:
            # The timed call as ONE captured hipGraph (dicp_amd.graphed.graphed_icp_step: call + loss + backward): what the headline shape does when the host is
            # the slow side -- an eager call is ~1 ms of host work against 2.7-3.3 ms of kernels (profiles/r06_host_time.txt), a replay is one launch.
            progress("leg: value_graphed")
            from dicp_amd.graphed import graphed_icp_step
            gi = new_icp(K)
            gs_, gt_ = data[0].detach().requires_grad_(True), data[1].detach().requires_grad_(True)
            gstep = graphed_icp_step(gi, lambda o_: o_["T"].sum(), gs_, gt_, T0, num_warmup_iters=4, trim_dist=TRIM, loss_fn=LOSS, dim=3)
            for _ in range(3):
                gout, ggr = gstep(gs_, gt_, T0)
            gtimes = []
            for _ in range(reps):
                fence()
                t0_ = time.perf_counter()
                gout, ggr = gstep(gs_, gt_, T0)
                fence()
                gtimes.append(time.perf_counter() - t0_)
            gstep.check_errors()
            extra["value_graphed"] = B * K / median(gtimes)
            extra["graphed_note"] = ("the timed call (K = %d, fwd + loss + bwd) captured once and replayed: %.3f ms per replay, every replay behind a synchronisation; results equal "
                                     "the eager call's (tests/test_gpu_configs.py::test_captured_step_after_a_synchronisation); finite: %s"
                                     % (K, median(gtimes) * 1e3, bool(torch.isfinite(gout["T"]).all() and torch.isfinite(ggr["source"]).all() and torch.isfinite(ggr["target"]).all())))
            del gi, gstep, gout, ggr, gs_, gt_
        if not brute:
            progress("leg: value_bruteforce (matrix cores, VALU)")
            for key, kv, what in (("value_bruteforce", L.KNN_MFMA, "the matrix-core brute force (split-f16 filter on v_mfma_f32_32x32x16_f16 + exact float32 refine)"),
                                  ("value_bruteforce_valu", L.KNN_VALU, "the float32 FMA brute-force kernel")):
                bf = new_icp(K, knn=kv)
                steady(bf, data, least=2, most=4)
                bt, _, _ = timed(bf, data, 3)
                extra[key] = world * B * K / median(bt)
                extra[key.replace("value_", "") + "_note"] = "same call with %s (all n*m pairs, same indices) in the loop: median of 3 calls, %.3f ms per step" % (what, median(bt) * 1e3 / K)
                del bf
        progress("leg: value_tolerance")
        tol = new_icp(50, tol=1e-4, const_iter=False)
        steady(tol, data, least=3, most=8)
        tt, (tout, _, _, _), _ = timed(tol, data, reps)
        k_exec = int(tout["deltas"].shape[1])
        extra["value_tolerance"] = world * B * k_exec / median(tt)
        extra["tolerance_note"] = ("tolerance 1e-4, max 50 iterations, const_iter off: %d iterations executed, %.3f ms per call, "
                                   "median of %d calls" % (k_exec, median(tt) * 1e3, reps))
        del tol, tout
        # LiDAR-like scenes: planar structure, two walls perpendicular to the sweep's sort axis
        progress("leg: value_structured")
        s2, t2 = make_scene_pairs(B, n, m, seed=3, dtype=torch.float32, first=rank * B)
        scene = (s2.to(dev), t2.to(dev))
        sc = new_icp(K)
        steady(sc, scene, least=3, most=8)
        st_, (sout, _, sgs, sgt), _ = timed(sc, scene, reps)
        sp, _ = sweep_stats(sc, K)
        extra["value_structured"] = world * B * K / median(st_)
        extra["structured_note"] = ("the same %d-iteration call on make_scene_pairs (ground plane 40 %%, four walls 12.5 %% each -- two of them perpendicular to x --, "
                                    "10 %% clutter; the search frame picks an oblique sort direction there, and the clouds' match certificates switch themselves off: "
                                    "8 %% of the queries sit within float32 rounding of a second candidate on the dense surfaces): %.3f ms per step, median of %d calls%s; finite: %s"
                                    % (K, median(st_) * 1e3 / K, reps, "" if sp is None else ", pairs scored %.2f %% of n*m per launch" % (100.0 * sp / (float(n) * m * B)),
                                       bool(torch.isfinite(sout["T"]).all() and torch.isfinite(sgs).all() and torch.isfinite(sgt).all())))
        if sp is not None:
            extra["structured_pairs_scored_fraction"] = sp / (float(n) * m * B)
        sane = sane and bool(torch.isfinite(sout["T"]).all() and torch.isfinite(sgs).all() and torch.isfinite(sgt).all())
        del sc, sout, sgs, sgt, scene, s2, t2
        # Inputs that are NOT the certificates' best case: source and target sampled independently from the same surfaces (no shared
        # point), 30 % of either cloud outside the other's footprint, 10 % clutter, start poses up to 0.2 rad / 1 m -- dense batches (the kernels'
        # own behaviour) and the same clouds as ragged lists (the reference's list inputs, tests/test_ICP_inputs.py:36-103: the host's list handling is in the call)
        def indep_leg(K_, const_iter, ragged, count):
            S, Tg = make_independent_pairs(B, n, m, seed=3, dtype=torch.float32, first=rank * B, ragged=bool(ragged))
            rows_kw = {}
            if ragged == "rows":        # the ragged clouds as ONE padded batch with per-cloud row counts (ICP.icp's source_rows / target_rows)
                real = float(sum(a.shape[0] * b.shape[0] for a, b in zip(S, Tg)))
                rows_kw = dict(source_rows=torch.tensor([x.shape[0] for x in S], dtype=torch.int32, device=dev),
                               target_rows=torch.tensor([x.shape[0] for x in Tg], dtype=torch.int32, device=dev))
                S = torch.nn.utils.rnn.pad_sequence(S, batch_first=True).to(dev)
                Tg = torch.nn.utils.rnn.pad_sequence(Tg, batch_first=True).to(dev)
                T0i, ragged = T0, False
            elif ragged:
                S, Tg = [x.to(dev) for x in S], [x.to(dev) for x in Tg]
                T0i = [torch.eye(4, device=dev)] * B
                real = float(sum(a.shape[0] * b.shape[0] for a, b in zip(S, Tg)))
            else:
                S, Tg, T0i = S.to(dev), Tg.to(dev), T0
                real = float(B) * n * m
            obj = new_icp(K_, tol=1e-12 if const_iter else 1e-4, const_iter=const_iter)

            def call():
                if ragged:
                    s_ = [x.detach().requires_grad_(True) for x in S]
                    t_ = [x.detach().requires_grad_(True) for x in Tg]
                else:
                    s_, t_ = S.detach().requires_grad_(True), Tg.detach().requires_grad_(True)
                o = obj.icp(s_, t_, T0i, trim_dist=TRIM, loss_fn=LOSS, dim=3, **rows_kw)
                o["T"].sum().backward()
                return o, s_, t_
            for _ in range(3):
                held = call()
            tms = []
            for _ in range(count):
                fence()
                t0_ = time.perf_counter()
                held = call()
                fence()
                tms.append(all_max(time.perf_counter() - t0_))
            o, s_, t_ = held
            k_exec = int(o["deltas"].shape[1])
            grads = [x.grad for x in (s_ + t_)] if ragged else [s_.grad, t_.grad]
            fin = bool(torch.isfinite(o["T"]).all()) and all(bool(torch.isfinite(g_).all()) for g_ in grads)
            real_pairs = real
            st = obj.knn_stats
            rec = {"cloud_it_per_s": world * B * k_exec / median(tms), "iterations_executed": k_exec, "ms_per_call": median(tms) * 1e3, "finite": fin,
                   "converged_clouds": int(o["stats"]["converged"].sum().item()) if "converged" in o["stats"] else None}
            if "knn_pairs" in st:
                rec["pairs_scored_fraction"] = float(st["knn_pairs"].sum().item()) / k_exec / real_pairs
            if "searched_again" in st:
                ag = st["searched_again"]
                rec["units_searched_again_by_iteration"] = [int(v) for v in ag[:k_exec, :64].sum(1).tolist()]
                rec["single_queries_searched_by_iteration"] = [int(v) for v in ag[:k_exec, 64:].sum(1).tolist()]
                rec["certs_off_clouds"] = int(st["certs_off"].sum().item()) if "certs_off" in st else None
            return rec, fin
        indep = {}
        for name, K_, ci, rg in (("k10", 10, True, False), ("tolerance", 50, False, False), ("k10_ragged_lists", 10, True, True), ("tolerance_ragged_lists", 50, False, True),
                                 ("k10_ragged_rows", 10, True, "rows")):
            progress("leg: value_independent / %s" % name)
            indep[name], fin_ = indep_leg(K_, ci, rg, 5)
            sane = sane and fin_
        extra["value_independent"] = indep["k10"]["cloud_it_per_s"]
        extra["independent"] = indep
        extra["independent_note"] = ("dicp_amd.synthetic.make_independent_pairs: source and target sampled INDEPENDENTLY from the same surfaces (corridor with partitions and pillars), "
                                     "footprints 6 m apart (30 %% of either cloud without counterpart), 10 %% clutter each, start poses up to 0.2 rad / 1 m; k10 = 10 constant "
                                     "iterations (compare value_k10 = %.0f on make_pairs, whose source IS target rows), tolerance = 1e-4 / max 50 / const_iter off; "
                                     "*_ragged_lists: the same clouds with lengths in [0.75, 1] x %d handed over as Python lists (the host's list handling is inside the call); k10_ragged_rows: "
                                     "those ragged clouds as ONE padded batch with per-cloud row counts (ICP.icp(source_rows=, target_rows=)): no lists, no 512 autograd leaves"
                                     % (extra.get("value_k10", float("nan")), n))
        if on_gpu and world == 1:
            progress("legs: configs[1], configs[3] slice and full batch, configs[0]")
            extra.update(other_configs(make_icp, dev, sync))
    gc.enable()

    # the pose all-gather on its own (N > 1): the collective bench.py overlaps with the backward, timed alone between fences
    progress("extra legs done")
    gather_ms = None
    if use_dist:
        gtimes = []
        for _ in range(5):
            fence()
            t0 = time.perf_counter()
            _, work = ddist.gather_poses_async(out["T"].detach(), total=B * abs(cw), force=True)
            if work is not None:
                work.wait()
            sync()
            gtimes.append(all_max(time.perf_counter() - t0))
        gather_ms = median(gtimes) * 1e3

    # three extra, untimed launches of the brute-force kNN kernel with HIP events: its roofline is reported
    # beside the running kernel's even when the (faster, exact) sweep kernel is the one in the loop
    bf = [float("nan")] * 3
    if on_gpu:
        tgt4 = _ops.pack_target(tgt)
        pose_id = torch.cat((torch.eye(3, device=dev).reshape(9), torch.zeros(3, device=dev))).repeat(B, 1).contiguous()
        idx_tmp = torch.empty((B, n), dtype=torch.int32, device=dev)
        bf = []
    bfm = []
    if on_gpu:
        img16 = _ops.f16_image(tgt4, m)
    for _ in range(3 if on_gpu else 0):
        for kv, dst in ((L.KNN_VALU, bf), (L.KNN_MFMA, bfm)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _ops.knn(src, pose_id, tgt4, m, kv, out=idx_tmp, image=img16)
            b.record()
            torch.cuda.synchronize()
            dst.append(a.elapsed_time(b))
    bf_ms = sorted(bf)[1] if on_gpu else None
    bfm_ms = sorted(bfm)[1] if on_gpu else None
    f16_again = (_ops.f16_counters(img16, B, tgt4.shape[1])[0] / 3.0 / (B * n)) if on_gpu else None

    # ---- the gate, on every rank: finite everywhere; rank 0 also holds its timed call against the oracle
    gate, base = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress("cpu baseline (bounded sample) and oracle gate")
        base = cpu_baseline(n, m)
        gate, ok = oracle_gate(src[:2], tgt[:2], K, out["T"][:2], gs[:2], gt[:2])
        sane = sane and ok
    if use_dist:
        flag = torch.tensor([1 if sane else 0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        sane = bool(flag.item())
    rc = 0 if sane else 1

    if rank == 0:
        def mean(v):
            return sum(v) / len(v) if v else None
        unit_w = True                                                    # run_call passes weight=None: the kernels read no weights (w_init == NULL)
        knn_ms, acc_ms = mean(ev_ms["knn"]), mean(ev_ms["accumulate"])
        # accumulate_bwd: the launches in which every cloud was at work (the truncated reverse sweep leaves the earlier iterations' launches
        # (next to) empty: their blocks read one flag and leave -- those are in the share of the call, not in the kernel's roofline)
        full = [v for k, v in enumerate(ev_ms["accumulate_bwd"]) if bwd_live is None or bwd_live[k] == B]
        bwd_ms = mean(full) if full else mean(ev_ms["accumulate_bwd"])
        flops_bf = 8.0 * n * m * B                                       # brute force, per launch
        flops = flops_bf if (brute or pairs_scored is None) else 8.0 * pairs_scored   # pairs the kernel actually scored
        knn_tf = flops / (knn_ms * 1e-3) / 1e12 if knn_ms else None
        acc_bytes = (44.0 if unit_w else 48.0) * n * B                   # per forward accumulate launch (4 B less: no w_init)
        bwd_bytes = (84.0 if unit_w else 88.0) * n * B                   # per backward launch (4 B less: no w_init)
        knn_traffic, knn_src = pmc_traffic("knn_valu" if brute else "knn_sweep", B, n)
        bwd_kernel = "accumulate_bwd_kernel" if brute else "accumulate_bwd_window_kernel"
        bwd_traffic, bwd_src = pmc_traffic(bwd_kernel, B, n)
        acc_traffic, acc_src = pmc_traffic("accumulate_kernel", B, n)
        bf_traffic, bf_src = pmc_traffic("knn_valu", B, n)
        bfm_traffic, bfm_src = pmc_traffic("knn_f16_kernel", B, n)
        last_ms = times[-1] * 1e3                                        # the call that carried the events
        share = {nm: (sum(v) / last_ms if v else 0.0) for nm, v in ev_ms.items()}
        # which scoring form each iteration's search took: the plan the object's earlier calls of this shape left (dicp_loop_buffers.search.form_plan; 0 = the default
        # by size, the matrix cores from 16384 targets on), the certifying search and the guard launches are vector code
        forms = None
        if on_gpu and not brute and certified is not None:
            try:
                rec = icp._hints.form_record(torch.device("cuda", torch.cuda.current_device()), (B, n, m, torch.float32))
                plan = rec.get("plan") or [0]
                cf_ = max(k for k in (icp._tuning.get("sweep_resort") or (0, 1, 2, 3)) if k < K)
                f16_default = m >= _ops.F16_SWEEP_MIN_TARGETS
                forms = [("matrix cores" if ((plan[min(k, len(plan) - 1)] == 2) or (plan[min(k, len(plan) - 1)] == 0 and f16_default)) else "vector") if k < cf_
                         else ("vector, certifying" if k == cf_ else "guard (vector)") for k in range(K)]
            except Exception:       # (a diagnostic: never the reason a line is lost)
                forms = None
        legs = {
            "knn": {"kernel": "knn (%s)" % ("brute force, " + args.knn if brute else "exact sorted sweep: same indices as brute force"),
                    "bound": "valu", "executes_on": ("valu f32 fma (same 157.3 TF peak as the f32 MFMA; the two share the ALUs)" if forms is None else
                                                      "per iteration as forms_by_iteration says: the plain searches of long slabs score on the matrix cores (knn_f16_sweep_kernel: "
                                                      "split-f16 filter on v_mfma_f32_32x32x16_f16, 32 executed f16 flop per pair, + exact float32 refine), the others on the "
                                                      "float32 vector ALUs; `achieved` counts SURVEY 8d's 8 flop per scored pair for either form against the float32 peak"),
                    "forms_by_iteration": forms,
                    "achieved": knn_tf, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": knn_tf / F32_PEAK_TFLOPS if knn_tf else None,
                    "traffic": knn_traffic, "traffic_source": knn_src, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC)",
                    "algorithmic_hbm_bytes": (16.0 * n + 16.0 * m) * B, "avg_launch_ms": knn_ms,
                    "flops_per_launch": flops,
                    "launch_ms_by_iteration": [round(v, 4) for v in ev_ms["knn"]],
                    "pairs_scored_fraction": None if (brute or pairs_scored is None) else pairs_scored / (float(n) * m * B),
                    "dense_equivalent_tflops": flops_bf / (knn_ms * 1e-3) / 1e12 if knn_ms else None,
                    "certified_iterations": certified,
                    "note": "8 flop per SCORED (query,target) pair vs the f32 peak, over ALL K search launches of a call (the certified iterations' "
                            "guard launches included; the rows their single-query searches score inside the accumulate launch are in the pair count, "
                            "<0.2 % of it); the kernel is FP32-compute-bound, its algorithmic HBM traffic is <1% of what HBM could move in its run time"},
            "accumulate": {"kernel": "accumulate (forward: residuals, weights, Jacobian, normal-equation sums)", "bound": "hbm",
                           "achieved": acc_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": acc_bytes / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if acc_ms else None,
                           "traffic": acc_traffic, "traffic_source": acc_src, "avg_launch_ms": acc_ms,
                           "algorithmic_bytes_per_launch": acc_bytes, "algorithmic_bytes_per_point": acc_bytes / (n * B),
                           "launch_ms_by_iteration": [round(v, 4) for v in ev_ms["accumulate"]]},
            "accumulate_bwd": {"kernel": bwd_kernel + (" (row atomics)" if brute else " (sorted space: LDS windows, no float atomics on the common path)"),
                               "bound": "hbm", "achieved": bwd_bytes / (bwd_ms * 1e-3) / 1e9 if bwd_ms else None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bwd_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bwd_ms else None,
                               "traffic": bwd_traffic, "traffic_source": bwd_src, "avg_launch_ms": bwd_ms,
                               "algorithmic_bytes_per_launch": bwd_bytes, "algorithmic_bytes_per_point": bwd_bytes / (n * B),
                               "launch_ms_by_iteration": [round(v, 4) for v in ev_ms["accumulate_bwd"]],
                               "clouds_at_work_by_iteration": bwd_live,
                               "avg_launch_ms_note": "mean over the launches in which all %d clouds were at work" % B if bwd_live is not None else None,
                               "truncated_reverse_sweep": None if bwd_live is None else
                               "a cloud's reverse sweep ends at the iteration from which on nothing -- that iteration's own contribution and the most any earlier one "
                               "could add given the recorded steps -- reaches 2^-22 of the cloud's largest contribution (below float32 rounding of the sums; the chain of "
                               "pose cotangents contracts by ~2e-4 per iteration near the pose: profiles/r03_cotangent_decay.txt); earlier iterations do no per-point "
                               "work.  Gradients equal the full sweep's to rounding: tests/test_gpu_skip.py, and every parity test runs with it on"},
        }
        for nm in legs:
            legs[nm]["share_of_timed_call"] = round(share[nm], 4)
            legs[nm]["total_ms_in_call"] = round(sum(ev_ms[nm]), 4) if ev_ms[nm] else None
        top = max(share, key=lambda nm: share[nm]) if any(share.values()) else "knn"
        line = {
            "metric": "ICP cloud-iterations/sec (fwd+bwd), B=%dx%d-pt clouds per GPU" % (B, n),
            "value": world * B * K / elapsed,
            "unit": "cloud-iterations/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "timed_calls": reps, "call_ms": [round(v * 1e3, 4) for v in times],
            "call_ms_by_rank": [round(v * 1e3, 4) for v in by_rank[i_med]],
            "pose_allgather_ms": gather_ms,
            "value_note": "median of %d timed %d-iteration calls (each: barrier + sync, icp() + backward(), barrier + sync; max over ranks; call_ms_by_rank: every "
                          "rank's own time of the median call); the last of them carries the HIP events of the roofline legs on its dispatches "
                          "(~6 %% slower for it: an event pair costs the queue ~4 us)" % (reps, K),
            "warmup_note": "W-iteration call, then %d untimed K-iteration calls (until three in a row agree to 3 %%: allocator + steady state at the timed shapes)" % steady_calls,
            "ms_per_step": elapsed * 1e3 / K,
            "batch_iterations_per_s": K / elapsed,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "ranks_seen": ranks_seen, "rccl_version": rccl, "head": head,
            "config": {"workload": "BASELINE configs[2]: B=%d/GPU synthetic %d-pt clouds, point-to-plane + huber(1.0) + "
                                   "trim(5.0), differentiable, dim=3, K=%d const iterations fwd + backward of T.sum() "
                                   "w.r.t. source and target" % (B, n, K),
                       "K": K, "clouds_per_gpu": B, "points": n, "icp_type": "pt2pl", "knn": args.knn,
                       "parallelism": "batch-sharded x%d, one pose all-gather per call" % world,
                       # (`value` grows with K -- the clouds converge in ~6 iterations, every further one is a certified iteration that costs a
                       #  sixth of an early one; the same run's other figures belong next to it wherever it is quoted)
                       "k_note": ("value is cloud-iterations/s at K = %d constant iterations, of which ~%d run after the clouds have converged; the same run at SURVEY 8d's "
                                  "K = 10: value_k10 = %s; in the reference's default tolerance mode (const_iter off, 6 iterations executed): value_tolerance = %s; on "
                                  "independently sampled, partially overlapping clouds with metre-sized start poses (K = 10): value_independent = %s"
                                  % (K, max(0, K - 6), ("%.0f" % extra["value_k10"]) if "value_k10" in extra else ("%.0f" % (world * B * K / elapsed) if K == 10 else "not run"),
                                     ("%.0f" % extra["value_tolerance"]) if "value_tolerance" in extra else "not run",
                                     ("%.0f" % extra["value_independent"]) if "value_independent" in extra else "not run"))},
            "roofline": dict(legs[top], dominant="largest share of the event-carrying timed call (%.3f ms): %s"
                             % (last_ms, ", ".join("%s %.0f %%" % (nm, 100 * share[nm]) for nm in sorted(share, key=lambda q: -share[q])))),
            "roofline_bruteforce_knn": {"kernel": "knn_valu_kernel (all n*m pairs, float32 FMA)", "bound": "valu",
                                        "achieved": flops_bf / (bf_ms * 1e-3) / 1e12 if bf_ms else None, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": flops_bf / (bf_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS if bf_ms else None, "traffic": bf_traffic, "traffic_source": bf_src,
                                        "avg_launch_ms": bf_ms, "measured": "3 extra launches outside the timed region, HIP events"},
            "roofline_bruteforce_knn_mfma": {
                "kernel": "knn_f16_kernel (all n*m pairs: split-f16 filter on v_mfma_f32_32x32x16_f16 + exact float32 refine; the same indices)", "bound": "mfma",
                "achieved": 4.0 * flops_bf / (bfm_ms * 1e-3) / 1e12 if bfm_ms else None, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": 4.0 * flops_bf / (bfm_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS if bfm_ms else None,
                "achieved_note": "EXECUTED f16 flops: one 32x32x16 MFMA (32768 flop) per 1024 pairs = 32 flop per pair, four times SURVEY 8d's algorithmic 8",
                "algorithmic_tflops": flops_bf / (bfm_ms * 1e-3) / 1e12 if bfm_ms else None,
                "algorithmic_frac_of_f32_peak": flops_bf / (bfm_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS if bfm_ms else None,
                "speedup_vs_valu_kernel": bf_ms / bfm_ms if (bf_ms and bfm_ms) else None,
                "avg_launch_ms": bfm_ms, "second_filter_pass_fraction_of_queries": f16_again,
                "traffic": bfm_traffic, "traffic_source": bfm_src, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC)",
                "bound_note": "the launch is bound by the VECTOR work beside the matrix pipe -- 8 v_min3 per MFMA for the lane-local minima + 3-4 of bookkeeping, and vector "
                              "and matrix instructions of one SIMD do not overlap here (scripts/ubench/mfma_valu_overlap.hip: 28 + 2.2 V cycles per MFMA with V vector "
                              "instructions) -- and by the clock the chip holds under it (1.7 GHz); matrix-pipe busy 48 % of the kernel here, 56 % at 256 x 65536 "
                              "(SQ_VALU_MFMA_BUSY_CYCLES, profiles/r04_knn_f16_c3_pmc.txt, profiles/r04_knn_c4_65536_pmc.txt)",
                "measured": "3 extra launches outside the timed region, HIP events"},
            "finite": sane,
        }
        if ev_ms["knn"] and ev_ms["accumulate"] and not brute:
            cf = max(k for k in (getattr(icp, "_tuning", {}).get("sweep_resort") or (0, 1, 2, 3)) if k < K) if certified is not None else K
            classes = {"full_search": list(range(0, min(cf, K))), "certifying_search": [cf] if cf < K else [], "certified": list(range(cf + 1, K))}
            line["ms_by_iteration_class"] = {
                nm: {"iterations": ks, "search_ms_mean": round(sum(ev_ms["knn"][k] for k in ks) / len(ks), 4),
                     "accumulate_ms_mean": round(sum(ev_ms["accumulate"][k] for k in ks) / len(ks), 4)} for nm, ks in classes.items() if ks}
            line["ms_by_iteration_class"]["note"] = ("HIP events of the last timed call; certified: the search launch is the guard (it works through the list of units the "
                                                     "previous step made: re-searches units with many spent budgets, single queries, re-scores candidate sets), the "
                                                     "accumulate streams each query's cached match row")
        for nm, key in (("knn", "roofline_knn"), ("accumulate", "roofline_accumulate"), ("accumulate_bwd", "roofline_streaming")):
            if nm != top:
                line[key] = legs[nm]
        line.update(extra)
        if base is not None:
            line["cpu_baseline"] = base
            line["check"] = gate
            line["speedup_vs_cpu"] = line["value"] / base["value"]
        if not sane:        # a wrong-result kernel must not emit a headline number
            line["invalid_value"] = line["value"]
            line["value"] = None
            line["error"] = "results not finite on some rank, or the timed call differs from the oracle (pose bar %g, gradient bar %g)" % (POSE_BAR, GRAD_BAR)
        (emit or (lambda text: os.write(_REAL_STDOUT, text.encode())))(json.dumps(line) + "\n")
    if use_dist:
        torch.distributed.destroy_process_group()
    return rc


if __name__ == "__main__":
    # stdout carries exactly ONE line, the JSON: RCCL prints its version banner to stdout when the process group comes up
    # (and libraries may log there too), so everything else this process writes to fd 1 goes to stderr instead
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    sys.exit(main())
