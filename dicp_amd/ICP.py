"""Host side of MI355X-native differentiable ICP: same call surface as the reference's
``dICP.ICP.ICP`` (/root/reference/dICP/ICP.py:14-47), with the per-iteration loop
(ICP.py:131-260) executed by libdicp_hip.so as one autograd node.

    icp = ICP(icp_type='pt2pl', differentiable=True, max_iterations=100, tolerance=1e-12)
    out = icp.icp(source, target, T_init, weight=None, trim_dist=5.0,
                  loss_fn={"name": "huber", "metric": 1.0}, dim=3)
    out["T"], out["pc"], out["deltas"], out["weights"], out["costs"], out["stats"]

Input forms, result keys / shapes / dtypes, config keys and error behaviour follow the
reference; citations are inline.  Inputs may live on the CPU (the reference's tests do):
they are moved to the visible HIP device for the computation and the results are returned
on the inputs' device.  Without a HIP device this class raises -- no CPU fallback exists.
"""
import os.path as osp

import torch
import yaml

from . import _call, _lib
from ._ops import F16_SWEEP_MIN_TARGETS, F16_SWEEP_STATIC_TARGETS, KabschLoop, compute_device, pack_list, packable, prebuild_search, transform_points
from ._loop import CallHints, ICPLoop, LoopConfig, form_tally_wanted, resort_schedule
from .nn import nn


class ICP:
    def __init__(self, config_path=None, icp_type='pt2pl', max_iterations=100, tolerance=1e-12, differentiable=True):
        # ICP.py:16-27 -- the YAML ships inside the package (the reference reads ../config/)
        if config_path is None:
            config_path = osp.join(osp.dirname(osp.abspath(__file__)), 'config', 'dICP_config.yaml')
        with open(config_path, 'r') as f:
            self.config = yaml.safe_load(f)

        # ICP.py:30-38 -- plain mutable attributes (tests poke them: test_ICP_inputs.py:263)
        prm, log, fun = (self.config['dICP'][k] for k in ('parameters', 'logging', 'functionality'))
        self.icp_type = icp_type
        self.max_iterations = max_iterations
        self.tolerance = tolerance
        self.const_iter = prm['const_iter']
        self.verbose = log['verbose']
        self.target_pad_val = prm['target_pad_val']
        self.source_zeroes_are_pad = prm['source_zeroes_are_pad']
        self.match_ratio_thresh = log['matched_ratio_thresh']
        self.diff = differentiable
        # ICP.py:40-44
        self.nn = nn(self.diff, use_gumbel=fun['gumbel'], eps=fun['gumbel_eps'], tau=fun['gumbel_tau'])
        # Build-specific attributes (not in the reference).  Every one leaves the RESULTS as they are; they choose among exact forms.
        self.knn_variant = _lib.KNN_AUTO      # which search the loop uses: _lib.KNN_AUTO | KNN_VALU | KNN_MFMA (brute force) | KNN_SWEEP (exact, slab-pruned)
        self.knn_stats = {}                   # OUTPUT: statistics of the last call ("knn_pairs": pairs scored by its sweep searches, device int64 shards: .sum(); ...)
        self.reuse_matches = True             # sweep path: match certificates -- search only where a match is not PROVEN unchanged since the last search
        self.bwd_window = True                # sweep path: backward in sorted space (LDS windows); False: row atomics
        # backward: an iteration whose normal-equation cotangent has decayed below this fraction of the cloud's largest adds nothing above
        # rounding and does no per-point work for that cloud (None: 2^-22 float32 / 2^-40 float64; 0: every iteration, like autograd)
        self.bwd_skip_eps = None
        # tolerance mode: iterations enqueued between two host checks of "all converged" (ICP.py:259).  None = auto:
        # every iteration for big batches (an iteration costs far more than a sync), every 4th for small ones
        # (converged clouds are frozen, so the extra iterations change nothing and the histories are trimmed)
        self.sync_every = None
        # True: a backward pass that used the one-launch tail of the truncated reverse sweep waits for itself and raises _loop.TailTimeout IN that pass if a
        # wait inside the launch ran out (a GPU kept full by other work for half a second) -- before the NaN gradients it would otherwise return reach an
        # optimizer.  False (default): no synchronisation; the failure is raised by the next backward pass of this object, or by check_errors().
        self.strict_errors = False
        # True: gradients are the same bits on every run.  By default a target row's contributions are summed in the order the GPU's waves reach it (and the
        # few whose match lies outside its block's window with float atomics), and the slot order of the backward comes from a counting sort that is not
        # reproducible inside a bucket: two runs of one call differ by parts in 1e4 of the largest gradient (float32, profiles/r04_soak.txt), within the
        # 1e-3 bar but unlike the reference, which is deterministic.  With the switch the call takes the exact sweep search and the windowed backward
        # (whatever knn_variant says), the backward's slot order is a stable sort, every window row is summed in slot order and the out-of-window rows are
        # added by a fixed-order launch instead of atomics; the one-launch tail and the one-call path are not used.  The forward's results do not depend
        # on it.  Cost: +0.7 ms per 10-iteration call at 256 x 16384, 2.4x the backward on clouds that do not converge (profiles/r05_deterministic_cost.txt).  Not available with Gumbel correspondences; pt2pt_dICP_SVD and the
        # standalone nn / loss operators are not covered (their target gradients are row atomics).
        self.deterministic = False
        # Private switches of single mechanisms, all on: what the tests flip to hold each mechanism to the path without it (and what the
        # measurements in DESIGN.md A/B'd).  Not part of the call surface.
        self._tuning = dict(
            small_loop=True,                  # small clouds: one block per cloud runs whole chunks of iterations
            sweep_resort=None,                # iterations at which the sweep re-orders its queries by x under the current pose (None: (0,1,2,3); small calls (0,1))
            cert_from=None,                   # iteration of the certifying search (None: the last re-ordering of the queries)
            cert_sets=True,                   # a match with a runner-up within rounding keeps a set of 4 candidate rows, re-scored per iteration
            cert_hint=True,                   # a shape whose clouds all switched their certificates off is searched plainly in the next calls
            cert_backoff=True,                # certificates are switched off per cloud, on device, where proving costs more than searching
            first_search=True,                # iteration 0's search is enqueued with the index build, before the loop state is prepared
            plan_call=True,                   # constant-iteration calls: every segment of the loop behind one library call (dicp_icp_forward_plan)
            bwd_tail=True,                    # the ended iterations of the truncated reverse sweep run as one launch
            one_call=True,                    # calls that need none of the loop's host decisions: one library call per direction (dicp_call_*)
            timing_events=None)               # measurement only: an object with .handles(K) -> 6 K HIP events the loop's launches carry (bench.py's EventLog)
        self._hints = CallHints()             # private: what this object's earlier calls tell later ones about time (per device, stream and shape)
        self._eye = {}                        # private: identity start poses of pt2pt_dICP_SVD by (batch size, dtype, device)

    def icp(self, source, target, T_init, weight=None, trim_dist=None, loss_fn=None, dim=3, source_rows=None, target_rows=None):
        return self.dICP(source, target, T_init, weight, trim_dist, loss_fn, dim, source_rows, target_rows)      # ICP.py:46-47

    def dICP(self, source, target, T_init, weight=None, trim_dist=None, loss_fn=None, dim=3, source_rows=None, target_rows=None):
        """Point-to-point / point-to-plane ICP on a batch of scan pairs (ICP.py:49-303).

        source : (n,3|6) | (N,n,3|6) | list of (n_i,3|6);  only xyz is used
        target : (m,3|6) | (N,m,3|6) | list of (m_i,3|6);  pt2pl needs the normals in 3:6
        T_init : (4,4) | (N,4,4) | list of (4,4)
        weight : None | (n) | (N,n) | list of (n_i)|None
        trim_dist : None or a distance; loss_fn : None or {"name": "huber"|"cauchy"|"trim", "metric": x}
        dim : 3, or 2 to optimise rotation about z and translation in x,y only
        source_rows, target_rows (build-specific, optional): (N) row counts of a PADDED batch (N,n,3|6) / (N,m,3|6) -- ragged clouds without Python lists: rows
                 past a cloud's count take no part (whatever they hold), as if the clouds had been given as a list of their own lengths.  A list of 256
                 clouds is 512 leaf tensors to autograd, ~8 ms of host time per call; the padded batch is one
        returns {"pc" (N,n,3), "T" (N,4,4), "costs" (N,K,1), "deltas" (N,K,6,1),
                 "weights" (N,K,n*r,1), "stats": {"converged","iterations","matched_ratio"}}
        """
        assert dim == 2 or dim == 3, "dim must be 2 or 3"                                # ICP.py:79
        # weight=None on tensor inputs: the weights are all 1 -- the loop is told so (w0 = None) instead of reading a tensor of ones
        unit_w = (weight is None and isinstance(source, torch.Tensor) and isinstance(target, torch.Tensor) and len(source) > 0 and len(target) > 0
                  and not self.source_zeroes_are_pad and not (self.nn.differentiable and self.nn.use_gumbel))
        source, target, T_init, w_pts, rows, per_cloud_w = self._batch(source, target, T_init, weight, unit_weights=unit_w)   # ICP.py:85
        # (with source_zeroes_are_pad the reference's (N,1) weight is multiplied into an (N,n) mask first, ICP.py:445-446: it is per point from there on)
        per_cloud_w = per_cloud_w and source.shape[1] > 1 and not self.source_zeroes_are_pad
        assert source.dtype == target.dtype == T_init.dtype                              # ICP.py:96
        if self.icp_type == 'pt2pl':
            assert target.shape[2] == 6                                                  # ICP.py:103
        else:
            target = target[:, :, :3]                                                    # ICP.py:105
        if loss_fn is not None and loss_fn['name'] not in ('huber', 'cauchy', 'trim'):
            raise ValueError("Invalid loss name: {}".format(loss_fn['name']))            # loss.py:19
        home = source.device
        dev = home if source.is_cuda else compute_device()
        source, target, T_init = (t.to(dev) for t in (source, target, T_init))
        w_pts = w_pts.to(dev) if w_pts is not None else None
        src_rows, tgt_rows = self._device_rows(rows, dev)
        if source_rows is not None or target_rows is not None:
            if rows is not None:
                raise ValueError("source_rows / target_rows describe a padded batch: not together with lists of clouds")
            src_rows, tgt_rows = self._given_rows(source_rows, source, "source_rows"), self._given_rows(target_rows, target, "target_rows")
            if src_rows is not None:      # the rows past a cloud's own carry weight zero, like the pads of a list (ICP.py:386-398)
                live = (torch.arange(source.shape[1], device=dev)[None, :] < src_rows[:, None]).to(source.dtype)
                w_pts = live if w_pts is None else w_pts * live

        if dim == 2:                                                                     # ICP.py:107-116
            # (the masks are made once per dtype and device: a host-to-device copy per call cannot be captured into a hipGraph -- the reference's own
            #  test pair is dim = 2 -- and is 20 us of a call that has 50 us of kernels)
            key = ("dim2", source.dtype, dev)
            if key not in self._eye:
                self._eye[key] = torch.tensor([1.0, 1.0, 0.0, 1.0, 1.0, 0.0], dtype=source.dtype, device=dev)
            keep_t = self._eye[key][:target.shape[2]]
            source = source * self._eye[key][:3]
            target = target * keep_t

        loss_name = loss_fn['name'] if loss_fn is not None else None     # 'trim' is a valid loss too (loss.py:15-16)
        # the target sort / index build of the sweep path goes to the GPU before the rest of this function's host work
        soft = bool(self.nn.differentiable and self.nn.use_gumbel)
        if not soft:
            target = target.contiguous()
            source = source.contiguous()
        wants_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (source, target, T_init, w_pts) if t is not None)
        first_search = bool(self._tuning["first_search"]) and self._tuning["timing_events"] is None and not (wants_grad and not self.bwd_window)
        deterministic = bool(self.deterministic) and wants_grad
        if deterministic and soft:
            raise NotImplementedError("ICP.deterministic: not with Gumbel-softmax correspondences (their adjoint adds to the target with float atomics)")
        cfg = LoopConfig(
            icp_type=self.icp_type, differentiable=bool(self.diff), max_iterations=int(self.max_iterations),
            tolerance=float(self.tolerance), trim_dist=trim_dist, loss_name=loss_name,
            loss_metric=float(loss_fn['metric']) if loss_fn is not None else 1.0, dim=dim,
            const_iter=bool(self.const_iter),
            tanh_steepness=float(self.config['dICP']['parameters']['tanh_steepness']),   # ICP.py:119
            match_ratio_thresh=float(self.match_ratio_thresh),
            knn_variant=(_lib.KNN_SWEEP | (self.knn_variant & 0xff00)) if deterministic else self.knn_variant, bwd_window=True if deterministic else bool(self.bwd_window),
            deterministic=deterministic, stats_out=self.knn_stats, hints=self._hints,
            sync_every=self.sync_every, timing_events=self._tuning["timing_events"], small_loop=False if deterministic else bool(self._tuning["small_loop"]),
            src_rows=src_rows, tgt_rows=tgt_rows, sweep_resort=resort_schedule(self._tuning["sweep_resort"], source.shape[0], source.shape[1], int(self.max_iterations), bool(self.reuse_matches), self._tuning["cert_from"]), reuse_matches=bool(self.reuse_matches), cert_from=self._tuning["cert_from"],
            bwd_skip_eps=self.bwd_skip_eps, cert_backoff=bool(self._tuning["cert_backoff"]), cert_sets=bool(self._tuning["cert_sets"]), cert_hint=bool(self._tuning["cert_hint"]),
            plan_call=bool(self._tuning["plan_call"]), bwd_tail=bool(self._tuning["bwd_tail"]), first_search=first_search, strict_errors=bool(self.strict_errors),
            # nn.py:14-16 via ICP.py:140: soft correspondences -- the same library loop with dicp_gumbel_nn in place of the search, the same one node
            gumbel=(self.nn.eps, self.nn.tau, getattr(self.nn, "_inject_U", None)) if soft else None)
        T_c = T_init.contiguous()
        if self._tuning["one_call"] and not deterministic and _call.eligible(cfg, source, target, T_c, w_pts, wants_grad):
            # a call that needs none of the loop's host decisions: one library call per direction on one allocation (dicp_call_forward / _backward)
            T, pc, deltas, weights, costs, converged, iterations, matched = _call.CallLoop.apply(source, target, T_c, w_pts, cfg)
        else:
            # the target sort / index build of the sweep path goes to the GPU before the rest of the host work
            # (... and the first search right behind it, unless the loop will want original indices -- the atomic backward -- or carries timing events)
            if not soft:
                rec = self._hints.form_record(dev, (source.shape[0], source.shape[1], target.shape[1], source.dtype)) if source.is_cuda else None
                cfg.prebuilt = prebuild_search(source, target, cfg.knn_variant, wants_grad and bool(cfg.bwd_window), T_init, src_rows, tgt_rows, first_search=first_search,
                                               tally=target.shape[1] < F16_SWEEP_STATIC_TARGETS and form_tally_wanted(rec, target.shape[1] >= F16_SWEEP_MIN_TARGETS))
            T, pc, deltas, weights, costs, converged, iterations, matched = ICPLoop.apply(source.contiguous(), target.contiguous(), T_init, w_pts, cfg)

        if per_cloud_w:
            # an (N,1) weight stays (N,1) in the reference, so its "matches at the start" (ICP.py:248,269: sum over dim 1 of
            # w_init > thresh) counts ONE per cloud where the kernels, handed the expanded (N,n) tensor, count n: same
            # numerator, the reference's denominator (the kernels' ratio is float32(int / int): the integer is recovered exactly)
            n_pts = float(source.shape[1])
            on = (w_pts[:, 0] > self.match_ratio_thresh) & (matched > 0)
            matched = torch.where(on, torch.round(matched * n_pts), matched)
        if self.verbose:                                                                 # ICP.py:262-264
            print("ICP converged in {} iterations".format(deltas.shape[1]))
            print("Final del_T_ts: {}".format(torch.linalg.norm(deltas[:, -1])))

        if self.icp_type == 'pt2pt':                                                     # ICP.py:164-165
            weights = weights.repeat_interleave(3, dim=2)
        results = {                                                                      # ICP.py:283-303
            "pc": pc,
            "T": T,
            "costs": costs.unsqueeze(-1),
            "deltas": deltas.unsqueeze(-1),
            "weights": weights.unsqueeze(-1),
            "stats": {"converged": converged, "iterations": iterations, "matched_ratio": matched},
        }
        if home != dev:
            results = {k: (v.to(home) if k != "stats" else {s: x.to(home) for s, x in v.items()})
                       for k, v in results.items()}
        return results

    def check_errors(self):
        """Wait for the backward passes this object has enqueued so far and raise if one of them reported a failure of its own (today: a wait of the
        one-launch tail that ran out, _loop.TailTimeout -- that pass's gradients are NaN).  Without this call the error is raised by the next backward pass
        -- or, with `strict_errors = True`, by the failing pass itself.  Call it before `optimizer.step()` when neither is acceptable."""
        self._hints.check(wait=True)

    def pt2pt_dICP_SVD(self, source, target, T_init, trim_dist=None, huber_delta=None, dim=3, weight=None):
        """SVD-based point-to-point ICP (reference: ICP.py:533-591, "not yet integrated").

        Same signature and return value as the reference -- (transformed source, T_ts) -- for a single pair
        (n,3|6), (m,3|6), (4,4); additionally accepts every batched / list form ``icp`` accepts and then
        returns (N,n,3), (N,4,4).  The step is the correct Kabsch solution C = U diag(1,1,det U det V) V^T,
        r = mu_t - C mu_s (the reference multiplies by V instead of V^T, ICP.py:566-570, which is only right
        for planar scenes; on those -- e.g. the bundled tests/data -- both reach the same pose).
        ``T_init`` is folded in exactly as the reference does (ICP.py:545-547,578): the points are not moved by it -- the
        search starts from the raw source and the result is T_ts = T_total @ T_init with T_total the transform that was
        found, the returned cloud is T_total applied to the source.  (``self.svd_seed_T_init = True``, build-specific,
        instead uses ``T_init`` as the starting pose of the search and returns the pose found.)
        Differences kept honest: ``trim_dist`` gates matches farther than that (the reference ignores it);
        ``huber_delta`` and ``dim`` are accepted and ignored, as there; ``weight`` (build-specific) gives per-point
        weights.  Stops when sum |T p - nn|^2 < tolerance (ICP.py:585).
        """
        single = not isinstance(source, list) and source is not None and source.dim() == 2
        s_b, t_b, T_b, w_pts, rows, _ = self._batch(source, target, T_init, weight)
        t_b = t_b[:, :, :3]                                                              # ICP.py:548
        assert s_b.dtype == t_b.dtype == T_b.dtype
        home = s_b.device
        dev = home if s_b.is_cuda else compute_device()
        s_b, t_b, T_b, w_pts = (t.to(dev) for t in (s_b, t_b, T_b, w_pts))
        src_rows, tgt_rows = self._device_rows(rows, dev)
        seed = bool(getattr(self, "svd_seed_T_init", False))
        if seed:
            T_start = T_b
        else:       # (the identity start is read, never written: one tensor per batch size serves every call)
            key = (int(T_b.shape[0]), T_b.dtype, dev)
            T_start = self._eye.get(key)
            if T_start is None:
                if len(self._eye) > 16:
                    self._eye.clear()
                T_start = self._eye[key] = torch.eye(4, dtype=T_b.dtype, device=dev).expand(T_b.shape[0], 4, 4).contiguous()
        if self._tuning["one_call"] and _call.kabsch_eligible(s_b, t_b, T_start, w_pts, bool(self.const_iter), self.knn_variant, src_rows, tgt_rows):
            # the whole loop, the pose and the transformed cloud (ICP.py:581) from one library call per direction (dicp_kabsch_call_*)
            T_found, pc, costs, iterations = _call.KabschCall.apply(s_b, t_b, T_start, w_pts, int(self.max_iterations), trim_dist)
        else:
            T_found, costs, iterations = KabschLoop.apply(s_b, t_b, T_start, w_pts, int(self.max_iterations), float(self.tolerance),
                                                          trim_dist, bool(self.const_iter), self.knn_variant, src_rows, tgt_rows, self.sync_every)
            pc = transform_points(s_b, T_found)                                          # ICP.py:581
        if self.verbose:                                                                 # ICP.py:588-589
            print("ICP converged in {} iterations".format(int(iterations.max().item()) - 1))
        T = T_found if seed else torch.matmul(T_found, T_b)                              # ICP.py:578
        self.svd_stats = {"costs": costs, "iterations": iterations}
        if home != dev:
            pc, T = pc.to(home), T.to(home)
        return (pc[0], T[0]) if single else (pc, T)

    # ------------------------------------------------------------------ batching
    def batch_size_handling(self, source, target, T_init=None, weight=None):
        """Normalise the accepted input forms to batched tensors (ICP.py:305-511):
        source (N,n_max,3) zero-padded, target (N,m_max,c) padded with max(source)*target_pad_val,
        T_init (N,4,4) or None, weights (N,n_max) -- repeated x3 along dim 1 for pt2pt (ICP.py:508-509)."""
        s, t, T, w, _, _ = self._batch(source, target, T_init, weight)
        if self.icp_type == 'pt2pt':
            w = w.repeat_interleave(3, dim=1)
        return s, t, T, w

    @staticmethod
    def _device_rows(rows, dev):
        """Per-cloud row counts of a ragged batch as (N) int32 device tensors for the kernels (None: dense batch).
        Source: the cloud's own length -- the rows behind it are the zero-weight pads of ICP.py:386-398, which contribute exactly
        nothing.  Target: own length + 1 where pad rows follow: they are copies of ONE far point (ICP.py:460,472-477), and
        one copy takes part exactly as all of them would (first of equals, like argmin)."""
        if rows is None:
            return None, None
        src_len, tgt_len, n_max, m_max = rows
        out = []
        for lens, full, extra in ((src_len, n_max, 0), (tgt_len, m_max, 1)):
            if lens is None or all(v == full for v in lens):
                out.append(None)
            else:
                out.append(torch.tensor([min(v + extra, full) for v in lens], dtype=torch.int32).to(dev))
        return out[0], out[1]

    @staticmethod
    def _given_rows(rows, batch, name):
        """Caller-given row counts of a padded batch -> (N) int32 on the batch's device, or None (every row takes part)."""
        if rows is None:
            return None
        r = torch.as_tensor(rows).to(device=batch.device, dtype=torch.int32).reshape(-1).contiguous()
        if r.numel() != batch.shape[0]:
            raise ValueError("%s: one count per cloud (%d), got %d" % (name, batch.shape[0], r.numel()))
        if not isinstance(rows, torch.Tensor) or not rows.is_cuda:          # (host values: checked here; device counts are the caller's word)
            host = torch.as_tensor(rows).reshape(-1)
            if int(host.min()) < 1 or int(host.max()) > batch.shape[1]:
                raise ValueError("%s: counts must lie in [1, %d]" % (name, batch.shape[1]))
        return r

    def _tensor_weight(self, w, source_b):
        """A caller-supplied weight TENSOR (the list form is cast item by item below).  The reference multiplies it into the
        residuals with torch broadcasting (ICP.py:169): a shape that does not broadcast against (N,n) raises there, a
        different float dtype is promoted.  The kernels read raw (N,n) buffers of the cloud dtype, so both are settled
        here: the same error for a wrong shape, a cast (differentiable) for a different dtype.  -> (weight (N,n), one weight per CLOUD was given)"""
        if not isinstance(w, torch.Tensor):
            raise TypeError("weight must be a tensor for a tensor source (got %s)" % (type(w),))
        want = tuple(source_b.shape[:2])
        per_cloud = False
        if tuple(w.shape) != want:
            # ICP.py:169 multiplies with broadcasting: after the row-count assert (ICP.py:326) the one other shape that
            # multiplies against the (N,n) trim / loss weights is one weight per cloud, (N,1) -- and only for pt2pl (pt2pt
            # repeats the weight x3 along dim 1 first, ICP.py:508-509, and (N,3) no longer broadcasts against (N,3n))
            if w.dim() == 2 and w.shape[0] == want[0] and w.shape[1] == 1 and (self.icp_type == 'pt2pl' or want[1] == 1):
                per_cloud = True
                w = w.expand(want)
            else:
                raise RuntimeError("The size of tensor weight %s must match the source points %s" % (tuple(w.shape), want))
        return (w if w.dtype == source_b.dtype else w.to(source_b.dtype)).contiguous(), per_cloud

    def _batch(self, source, target, T_init, weight, unit_weights=False):
        """As batch_size_handling, with ONE weight per point (what the kernels consume), and the clouds' own lengths:
        -> (source_b, target_b, T_b, w, rows, per_cloud); rows = None, or (source lengths | None, target lengths | None, n_max, m_max)
        when a list was padded; per_cloud: the caller gave ONE weight per cloud, (N,1) (the statistics count it once per cloud, ICP.py:248).
        unit_weights (tensor inputs, weight None): w is not built (None) -- the caller knows it is all ones."""
        if weight is not None:                                                           # ICP.py:321-326
            if isinstance(source, list):
                assert len(source) == len(weight), "weight must be list of same length as source"
            else:
                assert source.shape[0] == weight.shape[0], "weight must have same number of rows as source"

        # whole-input None / empty -> one phony pair with zero weight and identity T (ICP.py:328-346)
        if source is None or target is None or len(source) == 0 or len(target) == 0:
            f32 = dict(dtype=torch.float32, device="cpu")
            return (torch.zeros((1, 1, 3), **f32), torch.zeros((1, 1, 6), **f32),
                    torch.eye(4, **f32).unsqueeze(0), torch.zeros((1, 1), **f32), None, False)

        # dtype / device / column count come from the first non-empty target (ICP.py:347-358)
        dt, dev, cols = torch.float32, "cpu", None
        if isinstance(target, torch.Tensor) and target.dim() in (2, 3) and target.shape[1] > 0:
            # a batched tensor answers for itself (iterating over it would build one view per cloud: ~70 us of host time at 256)
            dt, dev = target.dtype, target.device
            cols = target.shape[1] if target.dim() == 2 else target.shape[2]
        else:
            for t_i in target:
                if t_i is not None and len(t_i) > 0:
                    dt, dev = t_i.dtype, t_i.device
                    cols = t_i.shape[0] if (not isinstance(target, list) and target.dim() == 2) else t_i.shape[1]
                    break
        opts = dict(dtype=dt, device=dev)

        src_len = tgt_len = None
        per_cloud = False
        # ---- source and per-point prior weights (ICP.py:360-446)
        if (isinstance(source, list) and (weight is None or all(w_i is None for w_i in weight)) and packable(source, (3, 6))
                and source[0].dtype == dt and source[0].device == torch.device(dev)):
            # device tensors of the batch's dtype, no empty cloud, no weights: the whole list in one launch (and its gradients in one) instead of one op per cloud
            src_len = [int(s_i.shape[0]) for s_i in source]
            source_b = pack_list(source, 3)
            w = (torch.arange(source_b.shape[1], device=dev)[None, :] < torch.tensor(src_len, device=dev)[:, None]).to(dt)
        elif isinstance(source, list):
            pts, pri = [], []
            for i, s_i in enumerate(source):
                if len(s_i) == 0:                                   # empty cloud: one zero point, zero weight
                    pts.append(torch.zeros((1, 3), **opts))
                    pri.append(torch.zeros((1,), **opts))
                    continue
                if i > 0 and (s_i.dim() != 2 or s_i.shape[1] not in (3, 6)):
                    raise ValueError("source list must contain (n x 3/6) tensors")
                w_i = torch.ones(s_i.shape[0], **opts)
                if weight is not None and weight[i] is not None:
                    assert len(weight[i]) == s_i.shape[0], "weight must have same number of rows as source"
                    w_i = weight[i] * w_i
                pts.append(s_i[:, :3])
                pri.append(w_i)
            # one padded copy for the whole list (the reference grows the batch item by item: O(N^2) copies)
            source_b = torch.nn.utils.rnn.pad_sequence(pts, batch_first=True).to(**opts)
            w = torch.nn.utils.rnn.pad_sequence(pri, batch_first=True).to(**opts)
            src_len = [int(p_i.shape[0]) for p_i in pts]
        elif source.dim() == 2 and source.shape[1] in (3, 6):
            source_b = source[:, :3].unsqueeze(0)
            w, per_cloud = ((None if unit_weights else torch.ones((1, source_b.shape[1]), **opts)), False) if weight is None else self._tensor_weight(weight.unsqueeze(0), source_b)
        elif source.dim() == 3 and source.shape[2] in (3, 6):
            source_b = source[:, :, :3]
            w, per_cloud = ((None if unit_weights else torch.ones(source_b.shape[:2], **opts)), False) if weight is None else self._tensor_weight(weight, source_b)
        else:
            raise ValueError("source must be (n x 3/6) or (N x n x 3/6) or list len(N) (n_N x 3/6)")

        if self.source_zeroes_are_pad:                                                   # ICP.py:445-446
            w = w * (torch.linalg.norm(source_b, dim=2) != 0.0).to(dt)

        # ---- target, padded with a value no source point can be nearest to (ICP.py:448-491)
        if isinstance(target, list) and packable(target, (cols,)) and target[0].dtype == dt and target[0].device == source_b.device:
            pad = (torch.max(source_b.detach()) * self.target_pad_val).to(dt)            # ICP.py:460 (a device scalar: no host round trip)
            tgt_len = [int(t_i.shape[0]) for t_i in target]
            target_b = pack_list(target, cols, pad)
        elif isinstance(target, list):
            pad = torch.max(source_b) * self.target_pad_val                              # ICP.py:460
            rows, dead = [], []
            for i, t_i in enumerate(target):
                if len(t_i) == 0:                                   # empty target: one zero row, cloud switched off
                    rows.append(torch.zeros((1, cols), **opts))
                    dead.append(i)
                    continue
                if i > 0 and (t_i.dim() != 2 or t_i.shape[1] != cols):
                    raise ValueError("target list must contain (m x 3/6) tensors. All tensors must have same number of columns")
                rows.append(t_i)
            tgt_len = [int(r_i.shape[0]) for r_i in rows]
            lens = torch.tensor(tgt_len, device=dev)
            target_b = torch.nn.utils.rnn.pad_sequence(rows, batch_first=True).to(**opts)
            real = (torch.arange(target_b.shape[1], device=dev)[None, :] < lens[:, None]).unsqueeze(-1)
            target_b = torch.where(real, target_b, pad * torch.ones((), **opts))
            if dead:
                live = torch.ones((w.shape[0], 1), **opts)
                live[dead] = 0.0
                w = w * live                                                             # ICP.py:456,467
        elif target.dim() == 2 and target.shape[1] in (3, 6):
            target_b = target.unsqueeze(0)
        elif target.dim() == 3 and target.shape[2] in (3, 6):
            target_b = target
        else:
            raise ValueError("target must be (m x 3/6) or (N x m x 3/6) or list len(N) (m_N x 3/6)")

        # ---- initial transforms (ICP.py:493-504)
        if T_init is None:
            T_b = None
        elif isinstance(T_init, list):
            T_b = torch.stack(T_init, dim=0)
        elif T_init.shape == (4, 4):
            T_b = T_init.unsqueeze(0)
        elif T_init.dim() == 3 and T_init.shape[1:] == (4, 4):
            T_b = T_init
        else:
            raise ValueError("T_init must be (4 x 4) or (N x 4 x 4) or list len(N) (4 x 4)")
        ragged = (src_len, tgt_len, int(source_b.shape[1]), int(target_b.shape[1])) if (src_len is not None or tgt_len is not None) else None
        return source_b, target_b, T_b, w, ragged, per_cloud
