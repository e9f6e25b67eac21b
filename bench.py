#!/usr/bin/env python3
"""Headline benchmark: differentiable ICP iterations (fwd+bwd) on B=256 x 16384-point clouds.

    python bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2] ("C3"): per GPU B=256 synthetic scan pairs, n=m=16384,
point-to-plane + Huber(1.0) + soft trim(5.0), differentiable, dim=3, float32, T_init=I.
One "step" = one ICP iteration over the whole batch, forward AND backward: the timed region is
ONE differentiable icp() call of K constant iterations, backward of T.sum() w.r.t. source and
target, and (N>1) the RCCL all-gather of the poses.  Inputs are resident in HBM when timing starts.

For N>1 launch with torch.distributed.run (one rank per GPU, weak scaling: 256 clouds per rank).
Rank 0 prints ONE JSON line.  `roofline` is the kNN kernel (the dominant one, FP32-compute-bound:
8*n*m flops per cloud-iteration against the 157.3 TF f32 peak) timed with HIP events on the
launch stream; `roofline_streaming` is the HBM-bound backward accumulate kernel.  `cpu_baseline`
is the CPU oracle (the reference's op sequence) timed on this host for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dicp_amd import dist as ddist                      # noqa: E402
from dicp_amd.ICP import ICP                            # noqa: E402
from dicp_amd.synthetic import make_pairs               # noqa: E402

STEADY_CALLS, STEADY_MAX = 3, 40   # untimed K-iteration calls before the timed one: at least / at most
F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak
HBM_PEAK_GBS = 8000.0        # HBM3E spec
LOSS = {"name": "huber", "metric": 1.0}
TRIM = 5.0


class EventLog:
    """HIP events around the kNN launch (forward) and the accumulate_bwd launch (backward) of every iteration.
    The loop runs inside libdicp_hip.so (dicp_icp_forward / _backward), so the library records them: it is handed
    the raw hipEvent_t handles of these torch events (same HIP runtime) -- on the stream the kernels run on.  The
    sweep and windowed-backward launches carry their pair on the dispatch itself (hipExtLaunchKernel start / stop
    events: the kernel's own begin / end timestamps, no barrier packets in the timed queue); the brute-force and
    atomic forms are bracketed by hipEventRecord."""

    def __init__(self):
        self.K, self.ev, self.arr = 0, None, None

    def handles(self, K):
        import ctypes
        if self.K != K:
            self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * K)]
            for e in self.ev:
                e.record()                       # materialises the underlying hipEvent_t
            self.arr = (ctypes.c_void_p * (4 * K))(*[e.cuda_event for e in self.ev])
            self.K = K
        return ctypes.cast(self.arr, ctypes.c_void_p)

    def all_ms(self, name):
        off = 0 if name == "knn" else 2
        return [self.ev[4 * k + off].elapsed_time(self.ev[4 * k + off + 1]) for k in range(self.K)]

    def mean_ms(self, name):
        ts = self.all_ms(name)
        return sum(ts) / len(ts) if ts else None


def pmc_traffic(kernel_prefix, B, n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_hbm_traffic.json, made by
    scripts/pmc_summary.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command).
    PMC counters cannot be read from inside the process, so this is null when no matching profile exists."""
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
            return sum(k["hbm_bytes"] * k["launches"] for k in hits) / tot, os.path.basename(f)
    return None, None


def run_call(icp, src, tgt, T0, world):
    """world: number of ranks, or -1 to run the collective even with a single rank (smoke test of the N>1 path)."""
    s = src.detach().requires_grad_(True)
    t = tgt.detach().requires_grad_(True)
    out = icp.icp(s, t, T0, trim_dist=TRIM, loss_fn=LOSS, dim=3)
    out["T"].sum().backward()
    # every rank holds the same number of clouds: the shard sizes are known, no size exchange (and no host sync) before the gather
    T_all = ddist.gather_poses(out["T"], total=src.shape[0] * abs(world), force=True) if world != 1 else out["T"].detach()
    return out, T_all, s.grad, t.grad


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
    times, T_ref = [], None
    t_start = time.time()
    while len(times) < 4 and (time.time() - t_start) < budget_s:
        s, t = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
        t0 = time.time()
        ref = O.icp_batched(s, t, T0, torch.ones(Bc, n), icp_type="pt2pl", differentiable=True, max_iterations=K,
                            tolerance=1e-12, trim_dist=TRIM, loss_fn=LOSS, dim=3, const_iter=True, tanh_steepness=5.0)
        ref["T"].sum().backward()
        times.append(time.time() - t0)
        T_ref = ref["T"].detach()
    best = sorted(times)[len(times) // 2]
    return {"value": Bc * K / best, "unit": "cloud-iterations/s", "cores": cores, "kind": "port",
            "sample": "%d clouds x %d iterations fwd+bwd, n=m=%d, float32, median of %d runs (%.2f s each)"
                      % (Bc, K, n, len(times), best)}, T_ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=16384, help="points per cloud (source and target)")
    ap.add_argument("--knn", choices=["auto", "sweep", "valu", "mfma"], default="auto",
                    help="auto/sweep: exact slab-pruned kNN (same indices as brute force); valu/mfma: brute-force kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # synthetic inputs are generated on the CPU: keep N ranks from oversubscribing the host's cores
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 1) // max(1, world))))
    # DICP_BENCH_FORCE_DIST=1 runs the distributed code path (RCCL init, barrier, pose all-gather, max-reduce)
    # even with one rank: how the N>1 path is smoke-tested on a 1-GPU box
    force_dist = os.environ.get("DICP_BENCH_FORCE_DIST", "0") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    if use_dist:
        torch.distributed.init_process_group("nccl", device_id=dev)     # RCCL

    B, n, m, K, W = args.batch, args.points, args.points, args.steps, args.warmup
    src, tgt = make_pairs(B, n, m, seed=3, dtype=torch.float32, first=rank * B)
    src, tgt = src.to(dev), tgt.to(dev)
    T0 = torch.eye(4, device=dev).repeat(B, 1, 1)

    cw = world if not (force_dist and world == 1) else -1
    icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=max(W, 1), tolerance=1e-12)
    icp.const_iter = True
    icp.knn_variant = {"auto": 0, "sweep": 3, "valu": 1, "mfma": 2}[args.knn]
    brute = args.knn in ("valu", "mfma")      # (auto also runs brute force when the clouds are small: settled after the timed call)
    if W > 0:
        run_call(icp, src, tgt, T0, cw)                                  # W untimed warm-up steps
    icp.max_iterations = K
    # still untimed: calls at the timed call's own shapes, so that the caching allocator already owns the K-sized
    # history / saved-index buffers (the first use of a new size is a synchronous hipMalloc) and the chip is in its
    # steady state -- back-to-back calls get faster for a while (6.8, 6.15, 6.1, 5.95, 5.9 ms: scripts/call_repeat_diag.py);
    # the number reported is what a training loop that calls icp() every step sees
    log = EventLog()
    log.handles(K)                          # create the HIP events now: not part of the timed workload
    icp._timing_events = None if os.environ.get("DICP_BENCH_NO_EVENTS") == "1" else log      # (experiment switch: what do the events cost?)
    # a generational GC pass over this process's heap takes tens of ms (10 steps take 7 ms): whether one lands inside
    # the timed call depends on the allocation count so far, i.e. on things as irrelevant as argv -> collect now and
    # pause the collector.  The collection goes BEFORE the steady-state calls: the first call after one is ~0.8 ms
    # slower (it frees the previous calls' graphs and the caching allocator re-splits its blocks).
    import gc
    gc.collect()
    gc.disable()
    steady_calls, recent = 0, []
    while steady_calls < STEADY_MAX:        # the events ride along: their first use costs the host too
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        run_call(icp, src, tgt, T0, cw)
        torch.cuda.synchronize()
        recent = (recent + [time.perf_counter() - t_w])[-3:]
        steady_calls += 1
        # steady = the last three calls within 3 % of each other (a fresh box needs more calls than a warm one: clocks,
        # first-touch of code objects and of the allocator's pools); every rank makes the same number of calls
        done = steady_calls >= STEADY_CALLS and max(recent) <= 1.03 * min(recent)
        if use_dist:
            flag = torch.tensor([1 if done else 0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            done = bool(flag.item())
        if done:
            break

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    out, T_all, gs, gt = run_call(icp, src, tgt, T0, cw)                 # exactly K steps
    t_host = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get("DICP_BENCH_DIAG") == "1":
        sys.stderr.write("timed call: host-return %.2f ms, done %.2f ms\n" % (t_host * 1e3, elapsed * 1e3))
        for _ in range(3):
            fence(); a = time.perf_counter(); run_call(icp, src, tgt, T0, cw); b = time.perf_counter(); fence()
            sys.stderr.write("  again: host-return %.2f ms, done %.2f ms\n" % ((b - a) * 1e3, (time.perf_counter() - a) * 1e3))
    gc.enable()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    knn_ms = log.mean_ms("knn")
    bwd_ms = log.mean_ms("accumulate_bwd")
    sane = bool(torch.isfinite(out["T"]).all() and torch.isfinite(gs).all() and torch.isfinite(gt).all())

    # one extra, untimed launch of the brute-force kNN kernel with HIP events: its roofline is reported
    # beside the running kernel's even when the (faster, exact) sweep kernel is the one in the loop
    from dicp_amd import _ops, _lib as L
    tgt4 = _ops.pack_target(tgt)
    pose_id = torch.cat((torch.eye(3, device=dev).reshape(9), torch.zeros(3, device=dev))).repeat(B, 1).contiguous()
    idx_tmp = torch.empty((B, n), dtype=torch.int32, device=dev)
    bf = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _ops.knn(src, pose_id, tgt4, m, L.KNN_MFMA if args.knn == "mfma" else L.KNN_VALU, out=idx_tmp)
        b.record()
        torch.cuda.synchronize()
        bf.append(a.elapsed_time(b))
    bf_ms = sorted(bf)[1]
    # knn=auto takes the brute-force kernel for small clouds (no sorted-sweep statistics then)
    from dicp_amd._ops import auto_knn_kind
    if args.knn == "auto" and auto_knn_kind(B, n, m) != L.KNN_SWEEP:
        brute = True
    pairs_scored = None if brute else float(icp.knn_stats["knn_pairs"].sum().item()) / K      # per launch

    if rank == 0:
        flops_bf = 8.0 * n * m * B                                       # brute force, per launch
        flops = flops_bf if brute else 8.0 * pairs_scored                # pairs the kernel actually scored
        knn_tf = flops / (knn_ms * 1e-3) / 1e12
        bwd_bytes = 88.0 * n * B                                         # per backward launch
        knn_traffic, knn_src = pmc_traffic("knn_valu" if brute else "knn_sweep", B, n)
        bwd_kernel = "accumulate_bwd_kernel" if brute else "accumulate_bwd_window_kernel"
        bwd_traffic, bwd_src = pmc_traffic(bwd_kernel, B, n)
        bf_traffic, bf_src = pmc_traffic("knn_valu", B, n)
        line = {
            "metric": "ICP cloud-iterations/sec (fwd+bwd), B=%dx%d-pt clouds per GPU" % (B, n),
            "value": world * B * K / elapsed,
            "unit": "cloud-iterations/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "warmup_note": "W-iteration call, then %d untimed K-iteration calls (until three in a row agree to 3 %%: allocator + steady state at the timed shapes)" % steady_calls,
            "ms_per_step": elapsed * 1e3 / K,
            "batch_iterations_per_s": K / elapsed,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: B=%d/GPU synthetic %d-pt clouds, point-to-plane + huber(1.0) + "
                                   "trim(5.0), differentiable, dim=3, %d const iterations fwd + backward of T.sum() "
                                   "w.r.t. source and target" % (B, n, K),
                       "clouds_per_gpu": B, "points": n, "icp_type": "pt2pl", "knn": args.knn,
                       "parallelism": "batch-sharded x%d, one pose all-gather per call" % world},
            "roofline": {"kernel": "knn (%s)" % ("brute force, " + args.knn if brute else "exact sorted sweep: same indices as brute force"),
                         "bound": "mfma", "achieved": knn_tf, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": knn_tf / F32_PEAK_TFLOPS,
                         "traffic": knn_traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, " + str(knn_src) + ")",
                         "algorithmic_hbm_bytes": (16.0 * n + 16.0 * m) * B, "avg_launch_ms": knn_ms,
                         "flops_per_launch": flops,
                         "launch_ms_by_iteration": [round(v, 4) for v in log.all_ms("knn")],
                         "pairs_scored_fraction": None if brute else pairs_scored / (float(n) * m * B),
                         "dense_equivalent_tflops": flops_bf / (knn_ms * 1e-3) / 1e12,
                         "note": "8 flop per scored (query,target) pair vs the f32 MFMA(=VALU) peak; the kernel is FP32-compute-bound, "
                                 "its algorithmic HBM traffic is <1% of what HBM could move in its run time"},
            "roofline_bruteforce_knn": {"kernel": "knn_%s_kernel (all n*m pairs)" % ("mfma" if args.knn == "mfma" else "valu"), "bound": "mfma",
                                        "achieved": flops_bf / (bf_ms * 1e-3) / 1e12, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": flops_bf / (bf_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, "traffic": bf_traffic,
                                        "avg_launch_ms": bf_ms, "measured": "3 extra launches outside the timed region, HIP events"},
            "roofline_streaming": {"kernel": bwd_kernel + (" (row atomics)" if brute else " (sorted space: LDS windows, per-block slabs, no float atomics)"),
                                   "bound": "hbm", "achieved": bwd_bytes / (bwd_ms * 1e-3) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bwd_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "traffic": bwd_traffic, "traffic_source": bwd_src, "avg_launch_ms": bwd_ms,
                                   "algorithmic_bytes_per_launch": bwd_bytes,
                                   "launch_ms_by_iteration": [round(v, 4) for v in log.all_ms("accumulate_bwd")]},
            "finite": sane,
        }
        if world == 1 and not args.no_cpu_baseline:
            base, T_ref = cpu_baseline(n, m)
            line["cpu_baseline"] = base
            # correctness gate: the same 2 clouds x 3 iterations on the GPU vs the oracle
            chk = ICP(icp_type="pt2pl", differentiable=True, max_iterations=3, tolerance=1e-12)
            chk.const_iter = True
            chk.knn_variant = icp.knn_variant
            o = chk.icp(src[:T_ref.shape[0]], tgt[:T_ref.shape[0]], T0[:T_ref.shape[0]], trim_dist=TRIM, loss_fn=LOSS, dim=3)
            line["check"] = {"pose_max_abs_diff_vs_oracle": float((o["T"].cpu() - T_ref).abs().max())}
            line["speedup_vs_cpu"] = line["value"] / base["value"]
        os.write(_REAL_STDOUT, (json.dumps(line) + "\n").encode())
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    # stdout carries exactly ONE line, the JSON: RCCL prints its version banner to stdout when the process group comes up
    # (and libraries may log there too), so everything else this process writes to fd 1 goes to stderr instead
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    main()
