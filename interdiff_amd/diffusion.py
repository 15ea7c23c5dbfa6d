"""Sampler seam: ``p_sample_loop(model, shape, noise=, clip_denoised=False, model_kwargs=, denoised_fn=)``
(diffusion/gaussian_diffusion.py:598-736) with the step update on ``interdiff_inpaint`` /
``interdiff_posterior_step``.

Host-side design (not a translation of the reference loop):
  * the seven fp64 coefficient tables are built once (numpy) and only the three per-step SCALARS the
    START_X / FIXED_SMALL path needs -- c1[t], c2[t], sigma[t] = exp(.5 logvar[t]) -- are handed to the
    kernel as fp32 arguments; the reference re-uploads whole tables 6x per step (:1620);
  * ``t`` tensors for all steps are materialised once; each carries ``host_value`` so the correction hook
    never syncs on ``t[0]``;
  * x0-inpainting, the posterior mean and the noise add are two elementwise launches (one when no hook);
  * per-step noise: ``step_noise`` = tensor [steps,...] / callable(i, x) for deterministic parity, else the
    in-kernel Philox generator keyed by (seed, loop index);
  * with the in-kernel generator and a graph-safe denoiser the plain step [denoiser forward -> inpaint+posterior ->
    advance] is captured ONCE into a hipGraph whose per-step scalars (c1, c2, sigma, t, loop index, seed) live in
    HBM, and replayed for every step on which the correction hook is inactive (989 of 1000); hook steps run
    eagerly between replays.  The two routes are bit-identical (tests/test_hip_parity.py).
Sharding: ``shard=(first_clip, total_clips)`` says that the batch handed in is clips [first, first + B) of a
larger batch that other ranks (or other calls) hold the rest of.  The in-kernel noise is then drawn at the WHOLE batch's Philox
counters (element offset first * C * T: the reference fills one ``randn_like`` tensor for the whole batch, :532) and the
feed-forward tile is picked from the whole batch's token rows, so the shard's result is bit-identical to the same clips of an
unsharded run -- whatever the route (eager, graph, chains).
Only the configuration the eval path uses is implemented (ModelMeanType.START_X, ModelVarType.FIXED_SMALL,
clip_denoised=False, identity timestep map); anything else raises NotImplementedError.
"""
import itertools
import math
import os
from types import SimpleNamespace
import numpy as np
import torch
from . import _lib


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.):
    """gaussian_diffusion.py:20-64."""
    n = num_diffusion_timesteps
    if schedule_name == 'linear':
        scale = scale_betas * 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if schedule_name == 'cosine':
        abar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)], dtype=np.float64)
    raise NotImplementedError('unknown beta schedule: %s' % schedule_name)


GRAPH_BLOCKS = (49, 7, 1)          # plain steps per captured hipGraph (49 = the gap between two correction steps)
SPLIT_MAX_BATCH = int(os.environ.get('INTERDIFF_SPLIT_MAX_BATCH', 128))     # plain steps of a batch of up to this many clips run as N_CHAINS independent chains (see _graph_loop); measured up to 128 (two chains 6 - 15 % faster than one at 40 .. 128 clips, tools/small_batch_ab.py)
N_CHAINS = int(os.environ.get('INTERDIFF_CHAINS', 2))
STAGGER_STEPS = int(os.environ.get('INTERDIFF_STAGGER', 0))      # > 0: the chains step through the whole loop on their own streams, chain c this many plain steps behind chain c - 1, the hook called per half batch (measured 5 % SLOWER at B = 16, equal at B = 32, tools/stagger_ab.py: not the default); 0: chains forked / joined inside every graph block, whole-batch hook steps
MAX_GRAPH_SHAPES = 8               # captured (shape, mask, cond) entries kept per denoiser before the cache is dropped wholesale
_UID = itertools.count(1)


def fresh_seed():
    """A 62-bit seed from torch's global CPU generator: what a caller who passes no ``seed`` gets, so that every sampling call
    draws its own per-step noise stream (the reference calls ``randn_like`` afresh in every step of every call,
    gaussian_diffusion.py:532) while ``torch.manual_seed`` still makes a whole run reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class GaussianDiffusion:
    """START_X / FIXED_SMALL diffusion (the reference's create_gaussian_diffusion configuration)."""

    def __init__(self, betas):
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        # fp32 per-step scalars exactly as `_extract_into_tensor(...).float()` then fp32 math would give them
        self._c1 = self.posterior_mean_coef1.astype(np.float32)
        self._c2 = self.posterior_mean_coef2.astype(np.float32)
        self._sigma = np.exp(np.float32(0.5) * self.posterior_log_variance_clipped.astype(np.float32)).astype(np.float32)
        self._t_cache = {}
        self._tables = {}
        self.fuse_plain_step = True          # plain steps of the graph route: posterior update inside the denoiser's last GEMM
        self.chain_plain_steps = os.environ.get('INTERDIFF_CHAIN_STEPS', '1') != '0'      # ... and, inside a captured run of plain steps, the next step's embedding in the same launch (csrc/tail_h2.h)
        self.split_chains = True             # ... and, for batches that do not fill the chip, as two independent half-batch chains
        self.split_min_rows = None           # ... when the batch has more token rows than this (None: the denoiser's one_chain_max_rows(), see _graph_loop)
        self.stagger_steps = STAGGER_STEPS   # ... which step through the WHOLE loop on their own streams, this many steps apart, when the hook can be called per half batch
        self._uid = next(_UID)               # names this schedule in the per-denoiser graph cache (never reused, unlike id())

    # ------------------------------------------------------------------ helpers
    def _timesteps(self, B, device):
        key = (B, str(device))
        if key not in self._t_cache:
            self._t_cache[key] = torch.arange(self.num_timesteps, device=device, dtype=torch.int64)[:, None].repeat(1, B).contiguous()
        return self._t_cache[key]

    def _table(self, device):
        """[steps,4] fp32 rows {c1, c2, sigma (0 at t=0), t/1000} for the device-parameterised step kernels."""
        key = str(device)
        if key not in self._tables:
            sig = self._sigma.copy()
            sig[0] = 0.0
            blend = (np.arange(self.num_timesteps, dtype=np.float32) / np.float32(1000)).astype(np.float32)
            self._tables[key] = torch.from_numpy(np.stack([self._c1, self._c2, sig, blend], axis=1).astype(np.float32)).contiguous().to(device)
        return self._tables[key]

    def _graph_loop(self, model, img, model_kwargs, denoised_fn, seed, todo, dump_steps, t_start, shard=None):
        lib = _lib.load()
        y = model_kwargs.get('y', {})
        B, dev = img.shape[0], img.device
        first, total = (0, B) if shard is None else shard
        per_clip = img.numel() // B
        elem0 = first * per_clip                    # position of this batch's x[0] inside the whole (possibly sharded) batch: the Philox counter base
        rows = total * img.shape[-1]                # the WHOLE batch's token rows: what every launch's feed-forward tile is picked by (MDM._pick_ffn_tile)
        table = self._table(dev)
        has_mask = 'inpainting_mask' in y and 'inpainted_motion' in y
        mu8 = gc = None
        if has_mask:
            m = y['inpainting_mask']
            assert img.shape == m.shape == y['inpainted_motion'].shape
            mu8, gc = (m if m.dtype == torch.uint8 else m.view(torch.uint8)).contiguous(), y['inpainted_motion'].contiguous()
        cond = y['cond']
        # The captured graphs read the sample's inputs from buffers this cache entry OWNS (x, gt, mask, cond): one capture per
        # (denoiser, shape) then serves every sample -- an eval loop or an autoregressive rollout feeds a new cond / gt per sample
        # and must not pay a re-capture (57 denoiser forwards) each time.
        # The cache lives ON the denoiser object: the captured graphs bake in the addresses of its arena, workspace and memory
        # context, so they must die with it (a cache keyed by id(model) would replay freed memory once the id is recycled).
        cache = model.__dict__.setdefault('_graph_cache', {})
        key = (self._uid, tuple(img.shape), has_mask, tuple(cond.shape), model.ffn_graph_key(rows) if hasattr(model, 'ffn_graph_key') else 0)    # the captured launches bake the feed-forward kernel choice in
        st = cache.get(key)
        if st is None:
            st = SimpleNamespace(x=torch.zeros_like(img), x0=torch.empty_like(img), ts=torch.zeros(B, dtype=torch.int64, device=dev),
                                 state=torch.zeros(8, dtype=torch.int64, device=dev), cond=torch.empty_like(cond, memory_format=torch.contiguous_format),
                                 gt=torch.empty_like(img) if has_mask else None,
                                 mask=torch.empty(img.shape, dtype=torch.uint8, device=dev) if has_mask else None, graphs={})
            st.kwargs = {'y': {'cond': st.cond}}              # what the captured denoiser calls see
            if len(cache) >= MAX_GRAPH_SHAPES:
                # drop this cache's graphs, the buffers IT allocated (x, x0, cond, chain and hook workspaces) and the entries of the denoiser's
                # per-shape pools that were created FOR these graphs (st.pool_keys, recorded below); pool entries that existed before -- a
                # graph captured elsewhere (bench.py, an integrator following INTEGRATION.md) may have baked their addresses in -- stay
                evicted = list(cache.values())
                cache.clear()
                if hasattr(model, 'forget_shape_buffers'):
                    for old in evicted:
                        model.forget_shape_buffers(getattr(old, 'pool_keys', ()))
            cache[key] = st
            fresh = True
            pools_before = None
        else:
            fresh = False
        st.cond.copy_(cond)
        if has_mask:
            st.gt.copy_(gc)
            st.mask.copy_(mu8)
        if fresh and hasattr(model, 'shape_buffer_keys'):
            pools_before = model.shape_buffer_keys()        # (taken before the first fold of this shape allocates its memory context)
        model.prepare_memory(st.cond)                   # once per sample, on the current stream (inside the caller's clock)
        rows_kw = {'batch_rows': rows} if getattr(model, 'accepts_batch_rows', False) else {}
        if fresh:
            model(st.x, st.ts, out=st.x0, **st.kwargs, **rows_kw)               # warm-up: workspaces, kernel attributes
            if getattr(model, 'supports_forward_step', False):
                # ... and the fused step's own instantiations (last GEMM with the update in its epilogue, QKV kernel with the sampler
                # bookkeeping): their FIRST launch must not happen inside a capture, where a launch error cannot be reported
                scratch = SimpleNamespace(x=st.x.clone(), ts=st.ts.clone(), state=torch.tensor([2, 0, 1, 0, 0, 0, 0, 0], dtype=torch.int64, device=dev))
                model.forward_step(scratch.x, scratch.ts, table, scratch.state, gt=st.gt, mask=st.mask, **st.kwargs, **rows_kw)
                if getattr(model, 'step_chaining', False):     # ... and the chained forms of the step tail (csrc/tail_h2.h)
                    scratch.state.copy_(torch.tensor([2, 0, 1, 0, 0, 0, 0, 0], dtype=torch.int64))
                    scratch.ts.copy_(st.ts)
                    model.forward_step(scratch.x, scratch.ts, table, scratch.state, gt=st.gt, mask=st.mask, embed_next=True, **st.kwargs, **rows_kw)
                    model.forward_step(scratch.x, scratch.ts, table, scratch.state, gt=st.gt, mask=st.mask, embed_ready=True, **st.kwargs, **rows_kw)
            torch.cuda.synchronize(dev)
            if pools_before is not None:              # what this cache entry made the denoiser allocate: released with the entry
                st.pool_keys = model.shape_buffer_keys() - pools_before

        def posterior(x, x0, g, mk, st):
            _lib.check(lib.interdiff_posterior_step_dev(_lib.dptr(x), _lib.dptr(x0), _lib.dptr(g, allow_none=True),
                                                        _lib.dptr(mk, allow_none=True), x.numel(), _lib.dptr(table), _lib.dptr(st.state),
                                                        _lib.dptr(st.ts), B, _lib.stream()), 'posterior_step_dev')

        fused = self.fuse_plain_step and getattr(model, 'supports_forward_step', False)
        # Chains: at <= 16 clips every kernel of a step is one partial wave of workgroups bounded by latency (operand round trips,
        # kernel boundaries), so the two halves of the batch, stepped as independent kernel chains on two branches of the SAME captured
        # graph, overlap each other's dead time.  Clips never interact in a plain step, the noise of a chain is drawn at the whole
        # batch's counters (state[6]), so the result is bit-identical to the single chain.  Hook steps stay whole-batch.  An odd batch
        # splits into parts that differ by one clip (more than two chains measured slower: 0.296 / 0.309 vs 0.281 ms per step with 3 / 4 at B = 16).
        nch = N_CHAINS
        split = (fused and self.split_chains and nch > 1 and 2 * nch <= B <= SPLIT_MAX_BATCH
                 and B * img.shape[-1] > ((model.one_chain_max_rows() if hasattr(model, 'one_chain_max_rows') else getattr(model, 'FFN16_MAX_ROWS', 0)) if self.split_min_rows is None else self.split_min_rows))      # smaller batches: launch-latency bound either way, and the feed-forward's 16-row grid already spans the chip (tools/small_batch_ab.py: equal at B = 8, one chain 7 % faster at B = 4)
        if split and not hasattr(st, 'chains'):
            st.chains = []
            for c in range(nch):
                start = c * (B // nch) + min(c, B % nch)            # balanced contiguous parts (sizes differ by at most one clip)
                h = B // nch + (1 if c < B % nch else 0)
                sl = slice(start, start + h)
                st.chains.append(SimpleNamespace(
                    sl=sl, x=st.x[sl], ts=st.ts[sl], gt=st.gt[sl] if has_mask else None, mask=st.mask[sl] if has_mask else None,
                    state=st.state if c == 0 else torch.zeros(8, dtype=torch.int64, device=dev),
                    cond=torch.empty(cond.shape[0], h, cond.shape[2], device=dev),
                    memctx=torch.empty(model.memctx_floats(h, cond.shape[0]) if getattr(model, 'accepts_batch_rows', False) else model.memctx_floats(h), dtype=torch.float32, device=dev),
                    ws=torch.empty(model.workspace_bytes(h, img.shape[-1]), dtype=torch.uint8, device=dev), stream=torch.cuda.Stream(dev)))
        if split:
            for ch in st.chains:                      # this sample's memory, folded per chain (its layout is per batch)
                ch.cond.copy_(st.cond[:, ch.sl])
                model.prepare_memory(ch.cond, into=ch.memctx)

        chain_steps = self.chain_plain_steps and fused and getattr(model, 'step_chaining', False)
        if st.__dict__.get('chain_steps') != chain_steps:      # captured launches bake it in
            st.graphs.clear()
            st.chain_steps = chain_steps

        def enqueue_plain(k):
            """k consecutive plain steps on the current (capturing) stream: every per-step scalar is read from HBM, so they fit any position.
            Inside such a run nothing touches x or the workspace between two steps, so step i's last launch also computes step i + 1's embedding
            (``chain_steps``: MDM.forward_step embed_next / embed_ready, csrc/tail_h2.h -- same bits, one launch and its boundary less per step)."""
            link = lambda i: dict(embed_ready=i > 0, embed_next=i + 1 < k) if chain_steps else {}
            if split:                       # fork: each chain runs its k steps on its own branch; join at the end
                cur = torch.cuda.current_stream()
                for ch in st.chains:
                    ch.stream.wait_stream(cur)
                    with torch.cuda.stream(ch.stream):
                        for i in range(k):
                            model.forward_step(ch.x, ch.ts, table, ch.state, gt=ch.gt, mask=ch.mask, memctx=ch.memctx, ws=ch.ws, batch_rows=rows, **link(i))
                for ch in st.chains:
                    cur.wait_stream(ch.stream)
            else:
                for i in range(k):
                    if fused:               # the update runs in the epilogue of the denoiser's last GEMM (same bits)
                        model.forward_step(st.x, st.ts, table, st.state, gt=st.gt, mask=st.mask, **link(i), **st.kwargs, **rows_kw)
                    else:
                        model(st.x, st.ts, out=st.x0, **st.kwargs, **rows_kw)
                        posterior(st.x, st.x0, st.gt, st.mask, st)

        def graph_of(k):
            """hipGraph of k consecutive plain steps."""
            if (k, fused, split) not in st.graphs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    enqueue_plain(k)
                st.graphs[(k, fused, split)] = g
            return st.graphs[(k, fused, split)]

        # Hook steps.  The gate of the correction hook is host-known (t <= 500 and t % 50 == 0, eval_smpl_short.py:85) and a hook that
        # reads its one per-call scalar on the device (HipCorrection.apply_dev: blend weight = table[state[0]][3]) launches the same
        # kernels with the same arguments at every timestep: a WHOLE hook step [denoiser forward -> inpaint -> hook -> posterior update] is
        # then one captured graph, and the 49 plain steps before it ride in the same graph -- a 50-step segment of the loop is one launch.
        # Same kernels in the same order as the eager hook step: same bits (tests).  Hooks without apply_dev, or with debug outputs
        # switched on, keep the eager path.
        # (the staggered per-chain form below calls the hook eagerly per half batch: it needs none of this -- no workspace, no warm-up call, no input copies)
        staggered = split and self.stagger_steps > 0 and B % nch == 0 and denoised_fn is not None and hasattr(denoised_fn, 'slice_kwargs')
        hook_dev = (not staggered and denoised_fn is not None and getattr(denoised_fn, 'graph_capturable', False) and getattr(denoised_fn, 'debug', None) is None
                    and getattr(denoised_fn, 'is_active', None) is not None and all(k in y for k in ('inpainted_motion', 'hand_pose', 'beta', 'obj_points'))
                    and os.environ.get('INTERDIFF_EAGER_HOOK') != '1')
        if hook_dev:
            hooks = st.__dict__.setdefault('hooks', {})
            hk = hooks.get(denoised_fn._uid)
            shapes = tuple(tuple(y[k].shape) for k in ('hand_pose', 'beta', 'obj_points'))
            if hk is None or hk.shapes != shapes:
                hk = SimpleNamespace(shapes=shapes, ws=denoised_fn.workspace_for(B, img.shape[-1]), fresh=True,
                                     y=dict(inpainted_motion=st.gt if has_mask else torch.empty_like(img),
                                            **{k: torch.empty(y[k].shape, dtype=torch.float32, device=dev) for k in ('hand_pose', 'beta', 'obj_points')}))
                hooks[denoised_fn._uid] = hk
                for stale in [gk for gk in st.graphs if isinstance(gk, tuple) and gk[0] == 'hook' and gk[1] == denoised_fn._uid]:
                    del st.graphs[stale]
            for k in ('hand_pose', 'beta', 'obj_points') + (() if has_mask else ('inpainted_motion',)):      # this sample's hook inputs, in buffers the graphs know
                hk.y[k].copy_(y[k])
            if hk.fresh:                            # first launches of the hook's kernels must not happen inside a capture
                denoised_fn.apply_dev(hk.y['inpainted_motion'].clone(), table, torch.zeros(8, dtype=torch.int64, device=dev), hk.y, hk.ws)
                torch.cuda.synchronize(dev)
                hk.fresh = False

        def hook_graph(k):
            """hipGraph of k plain steps followed by ONE hook step."""
            key = ('hook', denoised_fn._uid, k, fused, split)
            if key not in st.graphs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if k:
                        enqueue_plain(k)
                    model(st.x, st.ts, out=st.x0, **st.kwargs, **rows_kw)
                    if has_mask:
                        _lib.check(lib.interdiff_inpaint(_lib.dptr(st.x0), _lib.dptr(st.gt), _lib.dptr(st.mask), st.x0.numel(), _lib.stream()), 'inpaint')
                    denoised_fn.apply_dev(st.x0, table, st.state, hk.y, hk.ws)
                    posterior(st.x, st.x0, None, None, st)
                    if split:
                        for ch in st.chains[1:]:
                            ch.state[:6].copy_(st.state[:6])     # the whole-batch update advanced chain 0's state; the others follow
                st.graphs[key] = g
            return st.graphs[key]
        st.x.copy_(img)
        if elem0 % 4:                               # (chain offsets inside the batch may be odd when T % 4 != 0: the per-row form of the fused update takes any offset)
            raise ValueError('a shard must start at a multiple of 4 elements (C * T = %d per clip)' % per_clip)
        st.state.copy_(torch.tensor([t_start, 0, int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0, 0, elem0, 0], dtype=torch.int64))
        gate = getattr(denoised_fn, 'is_active', None)
        active = lambda i: denoised_fn is not None and (gate is None or gate(i))
        if staggered:
            return self._staggered_chains(model, st, table, model_kwargs, denoised_fn, active, seed, todo, dump_steps, t_start, has_mask, rows, elem0)
        if split:                                   # the other chains' states: the same schedule position, their x starts c chain-sizes in
            for c, ch in enumerate(st.chains[1:], 1):
                ch.state.copy_(torch.tensor([t_start, 0, int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0, 0, elem0 + ch.sl.start * per_clip, 0], dtype=torch.int64))
        st.ts.fill_(t_start)
        ts_all = self._timesteps(B, dev)
        dump, it, i, end = [], 0, t_start, t_start - todo
        while i > end:
            if active(i) and hook_dev:
                hook_graph(0).replay()
                k = 1
            elif active(i):
                # hook step, two-call form; its denoiser forward is replayed from a graph too (24 eager launches cost the host more
                # than the GPU needs to run them)
                if 'fwd' not in st.graphs:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        model(st.x, st.ts, out=st.x0, **st.kwargs, **rows_kw)
                    st.graphs['fwd'] = g
                st.graphs['fwd'].replay()
                x0 = st.x0
                if has_mask:
                    _lib.check(lib.interdiff_inpaint(_lib.dptr(x0), _lib.dptr(st.gt), _lib.dptr(st.mask), x0.numel(), _lib.stream()), 'inpaint')
                t = ts_all[i]
                t.host_value = i
                x0 = denoised_fn(x0, t, model_kwargs).contiguous()
                posterior(st.x, x0, None, None, st)
                if split:
                    for ch in st.chains[1:]:
                        ch.state[:6].copy_(st.state[:6])     # the whole-batch update advanced chain 0's state; the others follow
                k = 1
            else:
                # length of the plain run ahead (up to the next hook step / dump point / end), replayed in the
                # largest captured block sizes: 989 plain steps of a 1000-step sample take ~25 graph launches
                run = 1
                while i - run > end and not active(i - run) and not (dump_steps is not None and (it + run - 1) in dump_steps):
                    run += 1
                k = next(b for b in GRAPH_BLOCKS if b <= run)
                if (hook_dev and k == run and i - run > end and active(i - run)
                        and not (dump_steps is not None and (it + k - 1) in dump_steps)):      # the run ends right at a hook step: one graph for both
                    hook_graph(k).replay()
                    k += 1
                else:
                    graph_of(k).replay()
            i -= k
            it += k
            if dump_steps is not None and (it - 1) in dump_steps:
                dump.append(st.x.clone())
        return dump if dump_steps is not None else st.x.clone()

    def _staggered_chains(self, model, st, table, model_kwargs, denoised_fn, active, seed, todo, dump_steps, t_start, has_mask, rows, elem0):
        """The two-chain form taken to the whole loop: every half batch is stepped from the first to the last timestep on its OWN stream --
        plain steps as captured per-chain graphs, hook steps eagerly on the half batch (``denoised_fn.slice_kwargs``: every operand of
        the hook is per clip, eval_smpl_short.py:88-106) -- and chain c starts c x ``stagger_steps`` plain steps after chain 0.  The
        correction (VALU-bound contact scan, a 16-workgroup ObjProjector) of one chain then runs beside the plain steps (matrix pipe,
        latency) of the other instead of stopping the whole sample eleven times.  Clips never interact and every chain draws its
        noise at the whole batch's Philox counters (state[6]), so the sample is bit-identical to the joined form and to the eager loop."""
        lib = _lib.load()
        dev, B = st.x.device, st.x.shape[0]
        chains, h = st.chains, st.x.shape[0] // len(st.chains)
        end = t_start - todo
        # ---- the schedule, the same for every chain: plain runs in captured block sizes, hook steps, dump points
        prog, i, it = [], t_start, 0
        lag_at = self.stagger_steps if todo > self.stagger_steps else 0
        while i > end:
            if active(i):
                prog.append(('hook', i))
                k = 1
            else:
                run = 1
                while (i - run > end and not active(i - run) and not (dump_steps is not None and (it + run - 1) in dump_steps)
                       and it + run != lag_at):
                    run += 1
                k = next(b for b in GRAPH_BLOCKS if b <= run)
                prog.append(('plain', k))
            i -= k
            it += k
            if it == lag_at:
                prog.append(('lag',))              # chain c + 1 may start once chain c is here
            if dump_steps is not None and (it - 1) in dump_steps:
                prog.append(('dump',))
        n_lag = next((j + 1 for j, op in enumerate(prog) if op[0] == 'lag'), 1 if prog else 0)       # ops chain c + 1 trails chain c by (host enqueue order)
        # ---- per-chain graphs (captured before anything is enqueued: a capture synchronises the device)
        cur = torch.cuda.current_stream(dev)
        need_k = sorted({op[1] for op in prog if op[0] == 'plain'})
        need_fwd = any(op[0] == 'hook' for op in prog)
        for c, ch in enumerate(chains):
            if not hasattr(ch, 'x0'):
                ch.x0, ch.graphs = st.x0[ch.sl], {}
            for k in need_k:
                if k not in ch.graphs:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for _ in range(k):
                            model.forward_step(ch.x, ch.ts, table, ch.state, gt=ch.gt, mask=ch.mask, memctx=ch.memctx, ws=ch.ws, batch_rows=rows)
                    ch.graphs[k] = g
            if need_fwd and 'fwd' not in ch.graphs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    model(ch.x, ch.ts, out=ch.x0, memctx=ch.memctx, ws=ch.ws, batch_rows=rows)
                ch.graphs['fwd'] = g
        for c, ch in enumerate(chains):
            ch.state.copy_(torch.tensor([t_start, 0, int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0, 0, elem0 + c * ch.x.numel(), 0], dtype=torch.int64))
        st.ts.fill_(t_start)
        ts_all = self._timesteps(B, dev)
        kw = [denoised_fn.slice_kwargs(model_kwargs, ch.sl) for ch in chains] if need_fwd else None
        dumps = [torch.empty_like(st.x) for op in prog if op[0] == 'dump']

        def run_op(c, ch, op, n_dump):
            if op[0] == 'plain':
                ch.graphs[op[1]].replay()
            elif op[0] == 'hook':
                i = op[1]
                ch.graphs['fwd'].replay()
                if has_mask:
                    _lib.check(lib.interdiff_inpaint(_lib.dptr(ch.x0), _lib.dptr(ch.gt), _lib.dptr(ch.mask), ch.x0.numel(), _lib.stream()), 'inpaint')
                t = ts_all[i][ch.sl]
                t.host_value = i
                x0 = denoised_fn(ch.x0, t, kw[c]).contiguous()
                _lib.check(lib.interdiff_posterior_step_dev(_lib.dptr(ch.x), _lib.dptr(x0), None, None, ch.x.numel(), _lib.dptr(table),
                                                            _lib.dptr(ch.state), _lib.dptr(ch.ts), h, _lib.stream()), 'posterior_step_dev')
            elif op[0] == 'dump':
                dumps[n_dump][ch.sl].copy_(ch.x)
            elif op[0] == 'lag' and c + 1 < len(chains):
                ch.lag_event = torch.cuda.Event()
                ch.lag_event.record()

        # ---- enqueue: chain c runs op j - c * n_lag at host step j, so that chain c's 'lag' event exists before chain c + 1 waits for it
        n_dump = [0] * len(chains)
        for ch in chains:
            ch.stream.wait_stream(cur)
        for j in range(len(prog) + n_lag * (len(chains) - 1)):
            for c, ch in enumerate(chains):
                jj = j - c * n_lag
                if not 0 <= jj < len(prog):
                    continue
                with torch.cuda.stream(ch.stream):
                    if jj == 0 and c > 0 and getattr(chains[c - 1], 'lag_event', None) is not None:
                        ch.stream.wait_event(chains[c - 1].lag_event)
                    run_op(c, ch, prog[jj], n_dump[c])
                    if prog[jj][0] == 'dump':
                        n_dump[c] += 1
        for ch in chains:
            cur.wait_stream(ch.stream)
            ch.lag_event = None
        return dumps if dump_steps is not None else st.x.clone()

    def _step(self, model, img, x0_buf, i, it, t, model_kwargs, denoised_fn, noise_i, seed, elem0=0, rows_kw={}):
        lib = _lib.load()
        y = model_kwargs.get('y', {})
        x0 = model(img, t, **model_kwargs, **rows_kw)
        if 'inpainting_mask' in y and 'inpainted_motion' in y:
            m, g = y['inpainting_mask'], y['inpainted_motion']
            assert x0.shape == m.shape == g.shape
            mu8, gc = (m if m.dtype == torch.uint8 else m.view(torch.uint8)).contiguous(), g.contiguous()
            _lib.check(lib.interdiff_inpaint(_lib.dptr(x0, torch.float32), _lib.dptr(gc, torch.float32), _lib.dptr(mu8),
                                             x0.numel(), _lib.stream()), 'inpaint')
        if denoised_fn is not None:
            x0 = denoised_fn(x0, t, model_kwargs)
        sigma = 0.0 if i == 0 else float(self._sigma[i])
        _lib.check(lib.interdiff_posterior_step_at(_lib.dptr(img, torch.float32), _lib.dptr(x0, torch.float32),
                                                   _lib.dptr(noise_i, torch.float32, allow_none=True), img.numel(),
                                                   float(self._c1[i]), float(self._c2[i]), sigma, seed, it, elem0, _lib.stream()),
                   'posterior_step')
        return x0

    # ------------------------------------------------------------------ public surface
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False, skip_timesteps=0, init_image=None, randomize_class=False,
                      cond_fn_with_grad=False, dump_steps=None, const_noise=False, step_noise=None, seed=None, n_steps=None,
                      use_graph=True, first_t=None, shard=None):
        """Same keyword surface as the reference (:598-614).  Extra: ``step_noise`` (tensor [n,...] or callable
        (loop_index, x) -> tensor) for deterministic parity, ``seed`` for the in-kernel generator (None: a fresh one per call
        from torch's global generator, see ``fresh_seed``), ``n_steps`` to
        run only the first n iterations (t = T-1 .. T-n) -- used by the bench / short-chain parity tests, ``use_graph=False``
        to force the eager route, ``first_t`` to enter the schedule at that timestep with ``noise`` taken as x_{first_t}
        (a window of the loop for measurements; default T-1), ``shard=(first_clip, total_clips)``: the batch is clips
        [first, first + B) of a larger one -- noise counters and the feed-forward tile are the larger batch's (module docstring);
        ``noise`` / ``step_noise`` tensors, when given, are this shard's slices."""
        if clip_denoised:
            raise NotImplementedError('clip_denoised=True is not used on the eval path (eval_smpl_short.py:153)')
        if cond_fn is not None or skip_timesteps or init_image is not None or randomize_class or cond_fn_with_grad or const_noise:
            raise NotImplementedError('only noise=, denoised_fn=, model_kwargs=, dump_steps= are live (SURVEY.md §8(b))')
        if model_kwargs is None:
            model_kwargs = {}
        if seed is None:
            seed = fresh_seed()
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        if shard is not None:
            shard = (int(shard[0]), int(shard[1]))
            if not 0 <= shard[0] <= shard[0] + shape[0] <= shard[1]:
                raise ValueError('shard=(first_clip, total_clips) must contain this batch of %d clips' % shape[0])
        first, total = (0, shape[0]) if shard is None else shard
        per_clip = int(np.prod(shape[1:]))
        elem0 = first * per_clip
        if shard is not None and elem0 % 4:
            raise ValueError('a shard must start at a multiple of 4 elements (C * T = %d per clip)' % per_clip)
        if noise is not None:
            img = noise.clone().contiguous().float()             # NOT inpainted when given (:691-692)
        else:
            lib = _lib.load()
            img = torch.empty(*shape, dtype=torch.float32, device=device)
            _lib.check(lib.interdiff_randn_at(_lib.dptr(img), img.numel(), seed, 0xFFFFFFFF, elem0, _lib.stream()), 'randn')
            y = model_kwargs.get('y', {})
            if 'inpainting_mask' in y and 'inpainted_motion' in y:
                m = y['inpainting_mask']
                mu8, gc = (m if m.dtype == torch.uint8 else m.view(torch.uint8)).contiguous(), y['inpainted_motion'].contiguous()
                _lib.check(lib.interdiff_inpaint(_lib.dptr(img), _lib.dptr(gc), _lib.dptr(mu8), img.numel(), _lib.stream()), 'inpaint')
        t_first = self.num_timesteps - 1 if first_t is None else int(first_t)
        if not 0 <= t_first < self.num_timesteps:
            raise ValueError('first_t outside the schedule')
        todo = t_first + 1 if n_steps is None else min(int(n_steps), t_first + 1)
        if (step_noise is None and use_graph and getattr(model, 'graph_safe', False) and img.is_cuda
                and 'cond' in model_kwargs.get('y', {}) and os.environ.get('INTERDIFF_NO_GRAPH') != '1'):
            return self._graph_loop(model, img, model_kwargs, denoised_fn, seed, todo, dump_steps, t_first, shard)
        rows_kw = {'batch_rows': total * shape[-1]} if shard is not None and getattr(model, 'accepts_batch_rows', False) else {}
        ts = self._timesteps(shape[0], device)
        dump = []
        cond = model_kwargs.get('y', {}).get('cond') if isinstance(model_kwargs.get('y', None), dict) else None
        if cond is not None and hasattr(model, 'prepare_memory'):
            model.prepare_memory(cond)                  # once per sample, like the graph route (never trust a cached fold across samples)
        for it, i in enumerate(range(t_first, t_first - todo, -1)):
            t = ts[i]
            t.host_value = i
            if step_noise is None:
                nz = None
            elif callable(step_noise):
                nz = step_noise(it, img).contiguous()
            else:
                nz = step_noise[it]
            self._step(model, img, None, i, it, t, model_kwargs, denoised_fn, nz, seed, elem0, rows_kw)
            if dump_steps is not None and it in dump_steps:
                dump.append(img.clone())
        return dump if dump_steps is not None else img


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:64-114.  Betas are re-derived from the kept alphas_cumprod; the timestep map is the
    identity when every step is kept (the only configuration the eval path builds)."""

    def __init__(self, use_timesteps, betas):
        base = GaussianDiffusion(betas)
        use = set(use_timesteps)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, ac in enumerate(base.alphas_cumprod):
            if i in use:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        if self.timestep_map != list(range(len(betas))):
            raise NotImplementedError('timestep respacing is not used by the reference eval path')
        super().__init__(np.array(new_betas))


def create_gaussian_diffusion(noise_schedule='cosine', diffusion_steps=1000):
    """model/diffusion_smpl.py:251-284 with its fixed defaults (predict x_start, sigma_small, no respacing)."""
    betas = get_named_beta_schedule(noise_schedule, diffusion_steps, 1.)
    return SpacedDiffusion(range(diffusion_steps), betas)
