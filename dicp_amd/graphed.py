"""Fixed-shape ICP calls without the host in the loop: one hipGraph for the forward, one for the backward.

A mid-size call (BASELINE configs[1]: 32 clouds of 4096 points, 10 iterations) has ~0.55 ms of kernels and ~0.7 ms of host
work in front of them -- Python, ctypes, the allocator, the autograd engine (scripts/host_breakdown.py) -- and a single
65-point pair has 0.05 ms of kernels behind 0.4 ms of host.  A training loop calls the same shape again and again: everything
the host does per call can be done ONCE.  `graphed_icp` captures `ICP.icp` (constant-iteration mode: the tolerance mode's host
checks of "all converged", ICP.py:259, cannot be captured) and its backward with torch.cuda.make_graphed_callables; a call of the
returned function copies its arguments into the graphs' static inputs and replays; `graphed_icp_step` captures the call, a fixed
loss and the backward as ONE graph.  Measured (scripts/graphed_timing.py, profiles/r03_hipgraph_mid_size.txt), forward + backward:
configs[1] 0.76 -> 0.64 (two graphs, loss outside) / 0.55 ms (one graph), one 65-point pair 0.45 -> 0.32 / 0.28 ms, the benchmark
shape (GPU-bound) unchanged.  The forward results are those of the eager call bit for bit (the same kernels in the same order).

The usual rules of graphed callables apply: shapes, dtypes, requires_grad flags and every keyword are fixed at capture; the
outputs are STATIC tensors, overwritten by the next call (clone what must survive); Gumbel-softmax correspondences with in-kernel noise are
refused (their per-iteration seeds are drawn on the host and would be frozen into the graph); the truncated reverse sweep's one-launch
tail is placed once, from the warm-up calls' live counters, and stays there in every replay (a cloud that is still at work in it
is swept there: exact either way)."""
import contextlib
import gc

import torch

from .ICP import ICP


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection inside a capture: a collection that happens to run there may free device objects of earlier calls
    (events, pinned buffers), which the runtime refuses while a stream is capturing (the process aborts)."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

OUTPUTS = ("T", "pc", "deltas", "weights", "costs")


def _check_tail_word(word):
    """A replay cannot report a wait of the backward's one-launch tail that ran out the way an eager call does (no host code runs between the kernels): the
    error word of the captured pass is a static tensor of the graph -- ``call.check_errors()`` / ``step.check_errors()`` wait for the replays made so far and
    raise _loop.TailTimeout if the last one raised it (its gradients are NaN).  Call it before the optimizer's step.
    `word` is the CAPTURED pass's own (taken from the object's statistics right after the capture: a later eager call of the same object replaces the entry
    there); None: the captured pass has no one-launch tail, nothing to report."""
    from ._loop import TailTimeout
    if word is not None and int(word.item()) != 0:
        raise TailTimeout("dicp_amd: a wait of the captured backward pass's one-launch tail ran out in the last replay; its gradients are NaN")


def _capturable(icp, what):
    if not icp.const_iter:
        raise ValueError("%s needs ICP.const_iter = True: tolerance mode reads convergence counters on the host between iterations" % what)
    # Gumbel-softmax correspondences draw one noise seed per iteration on the HOST (torch's CPU generator) and hand them to the kernels by value:
    # a capture would freeze them, and every replay would draw the identical noise.  Injected noise tensors (nn._inject_U) are read in place.
    if icp.nn.differentiable and icp.nn.use_gumbel and getattr(icp.nn, "_inject_U", None) is None:
        raise ValueError("%s: the Gumbel-softmax correspondence draws its noise seeds on the host at every call; a captured graph would replay ONE "
                         "draw for ever.  Call ICP.icp eagerly, or inject the noise tensors (nn._inject_U)" % what)


def graphed_icp(icp: ICP, source, target, T_init, weight=None, num_warmup_iters=3, **icp_kwargs):
    """Capture ``icp.icp(source, target, T_init, weight=weight, **icp_kwargs)`` for tensors of these shapes / dtypes / requires_grad flags.

    Returns ``call(source, target, T_init[, weight]) -> dict`` with the keys T (N,4,4), pc (N,n,3) [differentiable],
    deltas, weights, costs [as ICP.icp returns them] -- static tensors.  ``source`` / ``target`` must be dense batches (N,n,3) /
    (N,m,3|6) on the GPU (lists are ragged: their shapes change from call to call)."""
    _capturable(icp, "graphed_icp")
    for t, nm in ((source, "source"), (target, "target"), (T_init, "T_init")) + (((weight, "weight"),) if weight is not None else ()):
        if not (torch.is_tensor(t) and t.is_cuda):
            raise ValueError("graphed_icp(%s): a CUDA tensor of the call's shape is needed for the capture" % nm)

    def fn(s, t, T0, *w):
        out = icp.icp(s, t, T0, weight=(w[0] if w else None), **icp_kwargs)
        return tuple(out[k] for k in OUTPUTS)

    sample = tuple(x.detach().clone().requires_grad_(x.requires_grad) for x in (source, target, T_init) + ((weight,) if weight is not None else ()))
    with _no_gc():
        graphed = torch.cuda.make_graphed_callables(fn, sample, num_warmup_iters=num_warmup_iters)

    word = icp.knn_stats.get("bwd_tail_error")      # (the last pass that ran was the capture's backward, if the inputs carry gradients at all)

    def call(s, t, T0, *w):
        return dict(zip(OUTPUTS, graphed(s, t, T0, *w)))
    call.check_errors = lambda: _check_tail_word(word)
    return call


def graphed_icp_step(icp: ICP, loss_of, source, target, T_init, weight=None, num_warmup_iters=3, **icp_kwargs):
    """The whole step -- ``out = icp.icp(...); loss_of(out).backward()`` -- as ONE hipGraph: for loops whose loss is a fixed function of the ICP's
    outputs (the benchmark's ``T.sum()``, a pose error against fixed ground truth held in a tensor the closure reads in place).

    Returns ``step(source, target, T_init[, weight]) -> (out, grads)``: ``out`` as `graphed_icp`, ``grads`` a dict of the gradients of the
    arguments that required grad at capture ("source", "target", "T_init", "weight") -- all static tensors, overwritten by the next step."""
    _capturable(icp, "graphed_icp_step")
    names = ("source", "target", "T_init") + (("weight",) if weight is not None else ())
    given = (source, target, T_init) + ((weight,) if weight is not None else ())
    for t, nm in zip(given, names):
        if not (torch.is_tensor(t) and t.is_cuda):
            raise ValueError("graphed_icp_step(%s): a CUDA tensor of the call's shape is needed for the capture" % nm)
    static = [x.detach().clone().requires_grad_(x.requires_grad) for x in given]

    def run():
        for x in static:
            x.grad = None
        out = icp.icp(static[0], static[1], static[2], weight=(static[3] if len(static) > 3 else None), **icp_kwargs)
        loss_of(out).backward()
        return out

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, int(num_warmup_iters))):
            run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with _no_gc():
        with torch.cuda.graph(graph):
            out = run()
    outs = {k: out[k] for k in OUTPUTS}
    grads = {nm: x.grad for nm, x in zip(names, static) if x.requires_grad}
    word = icp.knn_stats.get("bwd_tail_error")      # the captured pass's own error word (a static tensor of the graph), or None

    def step(*args):
        for dst, src in zip(static, args):
            if dst.data_ptr() != src.data_ptr():
                dst.detach().copy_(src.detach())
        graph.replay()
        return outs, grads
    step.check_errors = lambda: _check_tail_word(word)
    return step
