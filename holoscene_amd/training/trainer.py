"""Stage-1 training step -- the loop body of HoloSceneTrainRunner.run() (training/holoscene_train.py:332-428)
without its logging: zero_grad, forward, loss, backward, (gradient exchange), Adam, LR step.

No ``.item()`` inside the step (the reference pays ~25 device syncs per iteration for its grad-norm
print, holoscene_train.py:366-372); scalars are returned as device tensors.
"""
import contextlib
import os
import warnings

import torch

from ..model.loss import HoloSceneLoss, unit_cotangent
from ..model import network as _net
from ..model.network import HoloSceneNetwork
from ..utils.conf import Conf
from . import distributed as dist_util
from .optim import build_optimizer, build_scheduler


def stock_conf(num_rays=1024, S=128, d_out=32, num_levels=16, base_size=16, end_size=2048, logmap=19, beta=0.1, use_bg_reg=True,
               mlp_precision="fp32", learning_rate=5.0e-4, eikonal_mode="analytic"):
    """confs/replica/room_0/replica_room_0.conf with 'R rays x S samples' mapped as SURVEY D5:
    N_samples_eval=S, N_samples=S/2, N_samples_extra=S/4."""
    return Conf(
        train=Conf(learning_rate=learning_rate, lr_factor_for_grid=20.0, num_pixels=num_rays, add_objectvio_iter=25000, max_total_iters=200000,
                   stop_iter=100000, sched_decay_rate=0.1),
        loss=Conf(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, smooth_weight=0.005, depth_weight=0.5, normal_l1_weight=0.05,
                  normal_cos_weight=0.05, semantic_loss="torch.nn.MSELoss", use_obj_opacity=True, semantic_weight=5.0, reg_vio_weight=0.01,
                  bg_reg_weight=0.01, depth_type="marigold"),
        model=Conf(
            feature_vector_size=256, scene_bounding_sphere=1.0, use_bg_reg=use_bg_reg, render_bg_iter=10, mlp_precision=mlp_precision,
            eikonal_mode=eikonal_mode,
            implicit_network=Conf(d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], weight_norm=True,
                                  multires=6, inside_outside=True, use_grid_feature=True, divide_factor=1.0, sigmoid=10,
                                  color_grid_feature=True, num_levels=num_levels, base_size=base_size, end_size=end_size, logmap=logmap),
            rendering_network=Conf(mode="idr", d_in=9, d_out=3, dims=[256, 256], weight_norm=True, multires_view=4, multires_point=4,
                                   multires_normal=4),
            density=Conf(params_init=Conf(beta=beta), beta_min=0.0001),
            ray_sampler=Conf(near=0.0, N_samples=S // 2, N_samples_eval=S, N_samples_extra=S // 4, eps=0.1, beta_iters=10, max_total_iters=5)))


def benchmark_model_state(model, beta, seed=42):
    """Measurement state of SURVEY 8(d): reference init, then the lin0 columns geometric init zeroes are
    refilled with N(0, 1e-2) (leaves the dead-gradient state, quirk Q5) and beta is set explicitly."""
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        v = model.implicit_network.lin0.weight_v
        v[:, 3:] = (torch.randn(v[:, 3:].shape, generator=g) * 1e-2).to(v.device)
        model.density.beta.fill_(beta)


class Stage1Trainer:
    """One Stage-1 training iteration, three execution strategies (same arithmetic):

    * ``optimizer="torch"``: torch.optim.Adam + ExponentialLR exactly as the reference wires them (parity tests).
    * ``optimizer="flat"`` (default): parameters/gradients in flat buffers, fused Adam kernel (training/flat.py),
      hash-table gradients scattered straight into the flat gradient buffer.
    * ``graph=True`` (needs "flat"): everything after the sampler -- render, Eikonal set, loss, backward, Adam --
      is captured once per variant (background pass on/off, collision term on/off) as a HIP graph and replayed;
      only rays + the data-dependent sampler run eagerly.  ~1 100 kernel launches per iteration collapse into one
      graph launch, which is what removes the host-side launch gaps (45 % of the iteration before).
    """

    def __init__(self, conf, device="cuda", num_images=8, seed=42, world_size=1, rank=0, optimizer="flat", graph=False, zero1=True,
                 freeze_parameters=False, inject_draws=False, data_parallel=None, exchange=None, table_step=None, draw_in_graph=True):
        """data_parallel: run the gradient exchange (default: world_size > 1; True with a one-rank process group exercises the
        collectives' code path on a single GPU).  exchange: "overlap" (default; env HOLOSCENE_EXCHANGE) = ZeRO-1 per segment with
        each hash table's segment exchanged on a side stream as soon as its gradient is final, collectives captured inside the
        iteration graph (RCCL only); "serial" = the whole exchange after the backward pass, outside the graph.
        table_step: reduce-and-step of the hash tables (below); None = on unless HOLOSCENE_TABLE_STEP=0.  False gives the optimiser
        path every data-parallel rank runs (zero-fill, scatter into the gradient tables, Adam sweep) in a single process.
        draw_in_graph: train_step_resident() captures the batch draw inside the iteration's graph when the dataset offers scheduled_draw():
        True / "ahead" = iteration k's graph draws batch k + 1, its workgroups riding in the colour table's scatter launch late in the backward
        pass (off the next iteration's critical path); "head" = batch k drawn by iteration k's first launch; False = one launch in front of every
        replay (the A/Bs of profiles/r06)."""
        torch.manual_seed(seed)
        self.conf = conf
        self.device = torch.device(device)
        self.world_size, self.rank = world_size, rank
        self.dp = world_size > 1 if data_parallel is None else bool(data_parallel)
        self.model = HoloSceneNetwork(conf=conf.get_config("model"), graph_node_dict=None, num_images=num_images).to(self.device)
        self.loss = HoloSceneLoss(**conf.get_config("loss"))
        self.lr = conf.get_float("train.learning_rate")
        self.lr_factor = lr_factor = conf.get_float("train.lr_factor_for_grid", default=1.0)
        self.decay_rate = decay_rate = conf.get_float("train.sched_decay_rate", default=0.1)
        # nepochs * ds_len with ds_len = fix_length: decay_steps == max_total_iters (holoscene_train.py:110-116, 166-169)
        self.decay_steps = decay_steps = conf.get_int("train.max_total_iters", default=200000)
        self.flat = None
        if optimizer == "flat":
            from .flat import FlatAdam
            # ZeRO-1 under data parallelism: the hash tables become segments of their own, in the order in which the backward pass
            # finishes their gradients -- colour table (right after the appearance backward, 0.6-0.75 ms before the end of the
            # pass), SDF table (before the trunk's weight-gradient GEMMs): flat.py
            net = self.model.implicit_network
            early = None
            if self.dp and zero1 and getattr(net, "use_grid_feature", False):
                early = ([net.color_encoding.embeddings] if getattr(net, "color_grid_feature", False) else []) + [net.encoding.embeddings]
            self.flat = FlatAdam(self.model, self.lr, lr_factor, decay_rate, decay_steps, world_size=world_size, rank=rank,
                                 shard_moments=zero1 and world_size > 1, early_params=early)
            self.optimizer = self.scheduler = None
        elif optimizer == "torch":
            self.optimizer = build_optimizer(self.model, self.lr, lr_factor)
            self.scheduler = build_scheduler(self.optimizer, decay_rate, decay_steps)
        else:
            raise ValueError(optimizer)
        if graph and self.flat is None:
            raise ValueError("graph=True needs optimizer='flat'")
        if graph and getattr(self.loss, "end_step", -1) > 0:
            # loss.end_step > 0 makes the depth / normal weights a host-side function of the step (loss.py:322-323): a captured
            # graph would freeze them at their capture-time value.  No stock conf sets it; run those eagerly rather than wrongly.
            raise ValueError("graph=True cannot be combined with loss.end_step > 0 (per-step host-side loss weights); use graph=False")
        self.use_graph = graph
        self.freeze_parameters = freeze_parameters  # tests: compute gradients but skip the update
        self.inject_draws = inject_draws            # tests: explicit random draws live in the graph's static input block
        self.zero1 = zero1 and self.dp
        self._setup_exchange(exchange)
        self.add_objectvio_iter = conf.get_int("train.add_objectvio_iter", default=100000)
        self.iter_step = 0
        self._graphs = {}
        self.draw_in_graph = bool(draw_in_graph)
        self._draw_ahead = draw_in_graph in (True, "ahead")
        self._resident_blocks = {}      # id(dataset) -> (dataset, input block, gt block, ScheduledDraw): ONE static batch block for all graph variants
        self._capture_dataset = None    # the dataset whose scheduled draw the graph being captured takes in (train_step_resident)
        # Reduce-and-step (csrc/hash_encode.hip: k_hash_bin_step): in the variants of the whole-iteration graph in which every hash
        # table has ONE gradient producer (all but the background-patch iterations), the tables take their Adam step inside that
        # scatter's reduction; the gradient tables are then never zero-filled, written or read.  HOLOSCENE_TABLE_STEP=0: off (A/B).
        # _table_step_ok: variant -> decided by counting the producers in the variant's first (plain) warm-up pass.
        self._table_step = (graph and self.flat is not None and not self.dp and not freeze_parameters
                            and (os.environ.get("HOLOSCENE_TABLE_STEP", "1") != "0" if table_step is None else bool(table_step))
                            and self.flat.table_steps_supported()
                            and self._full_graph_ok())       # (only the whole-iteration graph ticks the optimiser before the backward pass)
        self._table_step_ok = {}
        self._table_producers = {}      # variant -> producers per table, counted in the variant's first (plain) pass
        self._pass_fused = False
        # same seed on every rank -> identical initial parameters; from here on every rank draws from its own stream (frames, ray
        # jitter, inverse-CDF draws, Eikonal points): identical draws across data-parallel ranks would correlate the sampling noise
        if world_size > 1:
            torch.manual_seed(seed + 7919 * (rank + 1))

    # ------------------------------------------------------------------ data-parallel exchange
    def _setup_exchange(self, mode):
        """Decide how the per-iteration exchange runs (class docstring of `exchange`).  Overlap needs ZeRO-1 over flat buffers, a
        second segment and a backend whose collectives are stream-ordered and graph-capturable (RCCL): it is probed once --
        a captured all-reduce on a forked side stream, replayed and checked, agreed on by all ranks -- and the trainer falls back
        to the serial form with a warning if this software stack refuses."""
        mode = mode or os.environ.get("HOLOSCENE_EXCHANGE", "overlap")
        if mode not in ("overlap", "serial"):
            raise ValueError(f"exchange must be 'overlap' or 'serial', not {mode!r}")
        self._overlap = False
        self._early_done = ()
        self._xchg_stream = None
        if not (self.dp and self.zero1 and self.flat is not None and len(self.flat.segments) > 1 and mode == "overlap"):
            return
        import torch.distributed as dist
        if not (dist.is_initialized() and dist.get_backend() == "nccl"):
            return
        self._xchg_stream = torch.cuda.Stream(self.device)
        warm = torch.zeros(1, device=self.device)
        dist.all_reduce(warm)            # an eager collective first: communicator set-up (allocations, handle exchange) must not
        torch.cuda.synchronize()         # happen inside the capture the probe is about to start
        ok = self._probe_captured_collective() if self.use_graph else True
        flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        self._overlap = bool(flag.item() > 0)
        if not self._overlap and self.rank == 0:
            warnings.warn("collectives could not be captured in a HIP graph on a side stream: exchanging serially after the graph")
        if self._overlap:
            from ..hashencoder.backend import ScatterWatch
            for t in self._early_tables():
                t._hs_scatter_watch = ScatterWatch()

    def _probe_captured_collective(self):
        import torch.distributed as dist
        try:
            t = torch.ones(1024, device=self.device)
            torch.cuda.synchronize()
            self._drain_collective_watchdog()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                cur = torch.cuda.current_stream()
                self._xchg_stream.wait_stream(cur)
                with torch.cuda.stream(self._xchg_stream):
                    dist.all_reduce(t)
                cur.wait_stream(self._xchg_stream)
            g.replay()
            torch.cuda.synchronize()
            return bool((t == float(dist.get_world_size())).all())
        except Exception:       # noqa: BLE001 -- any refusal means "serial"
            return False

    def _early_tables(self):
        """The tables that own segments 0 .. n-2 of the flat buffers (flat.py lays them out first, one segment each)."""
        return self.flat.params[:len(self.flat.segments) - 1]

    def _arm_early_exchange(self):
        """Start of an iteration body (before the forward pass, so that the producers of the tables' gradients are counted):
        advance the optimiser state once and ask to be told when each table's gradient is final."""
        self._early_done = ()
        if self._overlap and not self.freeze_parameters:
            self.flat.tick()
            for s_, t in enumerate(self._early_tables()):
                t._hs_scatter_watch.arm(lambda s_=s_: self._exchange_early_segment(s_))

    def _exchange_early_segment(self, s_):
        """Called by the last scatter into table s_ (autograd thread, the scatter's stream current): fork the exchange stream off
        that point and run the segment's reduce-scatter -> shard Adam -> all-gather there, under the rest of the backward
        pass.  Inside a capture this forks the graph; `_finish_exchange` joins it."""
        cur = torch.cuda.current_stream()
        self._xchg_stream.wait_stream(cur)
        with torch.cuda.stream(self._xchg_stream):
            dist_util.exchange_segment(self.flat, s_, self.world_size)
        self._early_done = self._early_done + (s_,)

    def _finish_exchange(self):
        """End of the backward pass: exchange what has not been exchanged yet, join the exchange stream."""
        dist_util.exchange_and_step_flat(self.flat, self.world_size, zero1=self.zero1, done=self._early_done)
        if self._early_done:
            torch.cuda.current_stream().wait_stream(self._xchg_stream)
        self._early_done = ()

    # ------------------------------------------------------------------ eager path
    def _exchange_and_step(self):
        if self.freeze_parameters:      # tests: gradients only -- no exchange, no Adam step, no step / learning-rate tick
            return
        if self.flat is not None:
            if self.dp:
                self._finish_exchange()
            else:
                self.flat.step()
                self._leave_tables_clean()
        else:
            if self.dp:
                dist_util.average_gradients(self.model.parameters(), self.world_size)
            self.optimizer.step()
            self.scheduler.step()

    def train_step(self, indices, model_input, ground_truth, rng=None, depths=None):
        """rng: explicit random draws (HoloSceneNetwork.forward's dict).  With graph=True they are normally a reason to run
        eagerly; a trainer built with inject_draws=True instead keeps them in the graph's static input block, so the captured
        whole-iteration graph itself can be driven by the reference's draws (parity tests).  depths: optional
        {"z_vals", "z_eik"[, "bg_z"]} that replace the sampler's results downstream of it (the sampler still runs)."""
        if self.use_graph and self.inject_draws:
            if rng is None or not self._full_graph_ok():
                raise ValueError("inject_draws=True needs explicit draws and the whole-iteration graph path")
            return self._train_step_full_graph(model_input, ground_truth, rng=rng, depths=depths)
        if self.use_graph and rng is None:
            return self._train_step_graph(indices, model_input, ground_truth)
        self.model.train()
        if self.flat is not None:
            self.flat.zero_grad()
            self._arm_early_exchange()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        out = self.model(model_input, indices, iter_step=self.iter_step, rng=rng)
        out["iter_step"] = self.iter_step
        loss_out = self.loss(out, ground_truth, call_reg=self.iter_step >= self.add_objectvio_iter)
        loss_out["loss"].backward()
        if self.flat is not None:
            self.flat.gather_grads()
        self._exchange_and_step()
        self.iter_step += 1
        return out, loss_out

    # ------------------------------------------------------------------ graph path
    def _graph_body(self, st, with_bg, call_reg):
        self.flat.zero_grad()
        self._arm_early_exchange()
        bg = st["bg"] if with_bg else None
        out = self.model.render(st["rays"], st["z_vals"], st["z_eik"], None, bg=bg)
        out["iter_step"] = 0
        loss_out = self.loss(out, st["gt"], call_reg=call_reg)
        loss_out["loss"].backward(gradient=unit_cotangent(loss_out["loss"].device))
        self.flat.gather_grads()
        self._update_in_body()
        return out, loss_out

    def _update_in_body(self):
        """Tail of a captured iteration body: the Adam step (single process), or -- when the collectives are capturable -- the
        whole data-parallel exchange; otherwise the exchange follows the replay (`_after_replay`)."""
        if self.freeze_parameters:
            return
        if not self.dp:
            self.flat.step()
            self._leave_tables_clean()
        elif self._overlap:
            self._finish_exchange()

    def _leave_tables_clean(self):
        """The passes that step the tables inside their scatters start from all-zero gradient tables without clearing them
        (FlatAdam.zero_grad(tables=False)) and leave them so; every other pass of such a trainer clears them at its end."""
        if self._table_step and not self._pass_fused:
            self.flat.clear_table_grads()
        self._pass_fused = False

    def _after_replay(self):
        if self.dp and not self._overlap and not self.freeze_parameters:
            dist_util.exchange_and_step_flat(self.flat, self.world_size, zero1=self.zero1)

    # ------------------------------------------------------------------ whole-iteration graph
    def _capture_mode(self):
        """Thread-local capture checking.  Other threads of this process make calls that are illegal for ANY thread while some
        stream captures in "global" mode and would invalidate the capture: a process group's watchdog polls the events of in-flight
        collectives (hipEventQuery), the refill threads of a device-resident dataset wait for `consumed` events and enqueue
        host->device copies (datasets/ring.py).  None of them touches the capturing stream, so the capturing thread is the right
        scope of the check."""
        return "thread_local"

    def _full_graph_ok(self):
        """Rays, sampler (device-side loop control), render, loss, backward and Adam in ONE graph: possible whenever the sampler
        can run without host syncs (ray_sampler.CONTROL == "device", bf16 fused SDF queries)."""
        sm = self.model.ray_sampler
        return hasattr(sm, "device_control_ok") and sm.device_control_ok(self.model) and sm.device_control_ok(self.model, 0)

    def _full_body(self, st, with_bg, call_reg):
        model = self.model
        tick = self.flat if (self.flat is not None and not self.freeze_parameters and (not self.dp or self._overlap)) else None
        # reduce-and-step: the first pass of a variant runs the plain way and counts the tables' gradient producers
        variant, counting = (with_bg, call_reg), False
        fused = self._table_step and tick is not None and self._table_step_ok.get(variant, False)
        if self._table_step and tick is not None and variant not in self._table_step_ok and not torch.cuda.is_current_stream_capturing():
            counting, _net._be.SCATTER_COUNTS = True, {}
        self._pass_fused = fused
        # fused: the 1.3 MB memset rides in the prologue launch below
        zero = self.flat.zero_grad(tables=not fused, defer=fused)
        self._arm_early_exchange()
        steps = self.flat.table_steps(producers=self._table_producers.get(variant)) if fused else contextlib.nullcontext()
        # entered with grad enabled: the renderer differentiates through beta and the normalised weights, the samplers detach them
        # one launch: beta, every weight-normalised matrix, the iteration's uniform draws, the optimiser tick (csrc/iter_ops.hip)
        # (the serial data-parallel exchange ticks for itself after the replay: training/distributed.py)
        sizes = None if "rng" in st else model.uniform_sizes(st["input"]["uv"].shape[1], with_bg)     # (the background patch's draws come from the same pool)
        # st["draw"] (train_step_resident): the batch is drawn and gathered into st["input"] / st["gt"] inside the graph -- "head": this iteration's, by
        # the prologue launch; "ahead": the NEXT iteration's, offered to the colour table's scatter launch of this backward pass
        sd = st.get("draw")
        rider = {"draw": sd, "done": False} if (sd is not None and self._draw_ahead) else None
        with _net.iteration_prologue(model, tick, sizes, zero=zero, draw=None if rider is not None else sd, draw_ahead=rider) as drawn:
            with torch.no_grad():
                if "rng" in st:     # injected draws (static tensors the caller overwrites before each replay)
                    rng = st["rng"]
                elif drawn is not None:
                    rng = drawn
                else:
                    rng = model.draw_uniforms(st["input"]["uv"].shape[1], st["input"]["uv"].device, with_bg)    # one generator launch per iteration
                rays = model.prepare_rays(st["input"], rng)
                z_vals, z_eik = model.sample(rays, rng)
                rounds = model.ray_sampler._rounds
                bg = model.prepare_background(st["input"], rng) if with_bg else None
                model.ray_sampler._rounds = rounds      # report the main pass, not the background patch
                sampled = {"z_vals": z_vals, "z_eik": z_eik, "bg_z": None if bg is None else bg["z_vals"]}
                if "depths" in st:  # the caller's depths replace the sampler's downstream of it
                    z_vals, z_eik = st["depths"]["z_vals"], st["depths"]["z_eik"]
                    if bg is not None and "bg_z" in st["depths"]:
                        bg["z_vals"] = st["depths"]["bg_z"]
            out = model.render(rays, z_vals, z_eik, None, rng=rng, bg=bg)
            out["sampled"] = sampled
        out["iter_step"] = 0
        loss_out = self.loss(out, st["gt"], call_reg=call_reg)
        with steps:
            loss_out["loss"].backward(gradient=unit_cotangent(loss_out["loss"].device))
        if rider is not None and not rider["done"]:     # no colour-table scatter in this pass took it along: a launch of its own
            sd.launch()
            rider["done"] = True
        self.flat.gather_grads()
        if not torch.cuda.is_current_stream_capturing():
            _net.assert_relays_consumed()       # (warm-up passes: a host-side look at two Python containers)
        if counting:
            seen, _net._be.SCATTER_COUNTS = _net._be.SCATTER_COUNTS, None
            views = [self.flat.flat_g[self.flat.offsets[i]:].data_ptr() for i in range(self.flat.n_tables)]
            # every table with at least one in-place producer (with two -- the background-patch iteration's geometry table -- the first
            # accumulates, the last steps: hsTableStep.prior) and no scatter into anything else
            self._table_step_ok[variant] = all(seen.get(v, 0) >= 1 for v in views) and set(seen) <= set(views)
            self._table_producers[variant] = [seen.get(v, 0) for v in views]
        self._update_in_body()
        if fused and not torch.cuda.is_current_stream_capturing():
            # (warm-up passes only: a host read) a gradient that reached a table past its scatter -- through autograd's accumulation --
            # would have missed the step and broken the all-zero contract of the next pass
            if bool(self.flat.flat_g[:self.flat.tables_end].any()):
                raise RuntimeError("reduce-and-step: a hash table received gradient outside its scatter; set HOLOSCENE_TABLE_STEP=0")
        return out, loss_out

    @staticmethod
    def _flatten_draws(rng, dev, n_extra):
        """The injected-draw dict as a flat {name: device tensor} block with shapes that do not depend on the realised sampler
        round count: only the first N_samples_extra entries of a permutation are ever used (ray_sampler.py:269), the patch origin
        becomes a float pair."""
        flat = {}
        for k, v in rng.items():
            if k == "bg":
                flat["bg"] = Stage1Trainer._flatten_draws(v, dev, n_extra)
            elif k == "bg_xy0":
                flat[k] = torch.as_tensor([float(a) for a in v], dtype=torch.float32).to(dev)
            elif k == "perm":
                flat[k] = torch.as_tensor(v)[:n_extra].to(dev).long().contiguous()
            else:
                flat[k] = torch.as_tensor(v).to(dev).contiguous()
        return flat

    @staticmethod
    def _copy_tree(dst, src):
        for k, v in src.items():
            if isinstance(v, dict):
                Stage1Trainer._copy_tree(dst[k], v)
            else:
                dst[k].copy_(v)

    def _warm_up(self, body):
        """Two eager passes of a to-be-captured body on a side stream (allocator, lazy kernel loading).  They run the full body
        including its Adam node; parameters, moments and the step / learning-rate state are restored afterwards, so warm-up
        is invisible to training: hsAdamState.step stays equal to iter_step (checkpoint.load_optimizer_state relies on that for
        the render_bg_iter / add_objectvio_iter gating after a resume)."""
        flat = self.flat
        snap = [t.clone() for t in (flat.flat_p, flat.flat_m, flat.flat_v, flat.state)]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        cur.wait_stream(side)
        for t, s0 in zip((flat.flat_p, flat.flat_m, flat.flat_v, flat.state), snap):
            t.copy_(s0)
        self.model.implicit_network.invalidate_packed_weights()
        torch.cuda.synchronize()
        _net._be._backend.scatter_workspaces_idle()      # the capture stream takes the scatter work spaces over without a clearing launch
        self._drain_collective_watchdog()

    def _drain_collective_watchdog(self):
        """Before a capture that will pull the process group's communication stream into capture mode: wait until the process group's
        watchdog thread has RETIRED every eager collective it still tracks.  It polls their end events (hipEventQuery, every ~100 ms); on
        this stack a query of an event whose stream has meanwhile entered capture fails with hipErrorCapturedEvent, the watchdog rethrows
        and the process aborts (seen once in five runs of tests/test_distributed_gpu.py behind another GPU test: 'operation not permitted
        on an event last recorded in a capturing stream' from ProcessGroupNCCL::Watchdog::runLoop).  The warm-up passes' collectives have
        completed by now (synchronize above); what is left is the watchdog noticing.  That is observable: the process group's flight
        recorder marks an entry `retired` when the watchdog drops it from its list (tools/exp/pg_retire_probe.py: 60-100 ms after
        completion), so the wait is on that CONDITION -- the short sleeps only yield the GIL between polls.  Without a recorder
        (TORCH_FR_BUFFER_SIZE=0 when the group was created: no entries) three polling periods of sleep remain the fallback."""
        if not self.dp:
            return
        import time
        import torch.distributed as dist
        if not (dist.is_initialized() and dist.get_backend() == "nccl"):
            return
        try:
            import pickle
            from torch._C._distributed_c10d import _dump_nccl_trace
            deadline = time.monotonic() + 10.0
            while time.monotonic() < deadline:
                entries = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries", [])
                if not entries:
                    break                                   # recorder off: fall back
                if all(e.get("retired", False) for e in entries):
                    return
                time.sleep(0.005)
        except Exception:                                   # no flight recorder in this build
            pass
        time.sleep(0.35)

    def _train_step_full_graph(self, model_input, ground_truth, rng=None, depths=None):
        self.model.train()
        with_bg = self.model.wants_background(self.iter_step)
        base = ("full", with_bg, self.iter_step >= self.add_objectvio_iter)
        ds = self._capture_dataset if (rng is None and depths is None) else None      # train_step_resident: a graph that draws its own batches is wanted
        entry = self._sched_entry(base, ds) if ds is not None else None
        if entry is None:
            entry = self._graphs.get(base)
        if rng is not None:
            rng = self._flatten_draws(rng, self.device, self.model.ray_sampler.N_samples_extra)
        if entry is None:
            shared = self._resident_blocks.get(id(ds)) if ds is not None else None
            if shared is not None:      # every graph variant of a dataset reads ONE static batch block (draw-ahead: variant A's graph draws variant B's batch)
                st = {"input": shared[1], "gt": shared[2]}
            else:
                st = {"input": {k: v.clone() for k, v in model_input.items()}, "gt": {k: v.clone() for k, v in ground_truth.items()}}
            if rng is not None:
                st["rng"] = rng       # freshly made device tensors: they become the static block
            if depths is not None:
                st["depths"] = {k: v.to(self.device).clone() for k, v in depths.items()}
            sd = None
            if shared is not None:
                sd = shared[3]
            elif ds is not None:
                sd = ds.scheduled_draw(st["input"], st["gt"])      # None: this dataset cannot (the batch stays a launch of its own)
                if sd is not None:
                    self._resident_blocks[id(ds)] = (ds, st["input"], st["gt"], sd)
            if sd is not None:
                st["draw"] = sd
                # the warm-up passes below launch the draw eagerly: ring and cursor must be in place (ahead: and the block hold this batch)
                sd.before_replay_ahead() if self._draw_ahead else sd.before_replay()
            self._warm_up(lambda: self._full_body(st, base[1], base[2]))
            if sd is not None:
                sd.resync()                 # (they advanced the device's batch number: the next before_replay() puts it back)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=self._capture_mode()):
                out, loss_out = self._full_body(st, base[1], base[2])
            entry = {"graph": g, "static": st, "out": out, "loss": loss_out, "rounds": self.model.ray_sampler._rounds, "sched": sd}
            self._graphs[base + ("sched",) if sd is not None else base] = entry
        st = entry["static"]
        sd = entry.get("sched")
        if sd is not None:      # the graph draws its own batches: the one the caller peeked at (same batch number, same frame) is the one it trains on
            sd.before_replay_ahead() if self._draw_ahead else sd.before_replay()
        else:
            dl = [st["input"][k] for k in model_input] + [st["gt"][k] for k in ground_truth]
            sl = list(model_input.values()) + list(ground_truth.values())
            torch._foreach_copy_(dl, sl)
            if ds is not None:  # (peeked, and no graph of its own to draw it: consumed here)
                ds.next_batch()
        if rng is not None and st["rng"] is not rng:
            self._copy_tree(st["rng"], rng)
        if depths is not None:
            self._copy_tree(st["depths"], {k: v.to(self.device) for k, v in depths.items()})
        entry["graph"].replay()
        if sd is not None:
            sd.after_replay_ahead() if self._draw_ahead else sd.after_replay()
        self.model.ray_sampler._rounds = entry["rounds"]
        self._after_replay()
        self.iter_step += 1
        return entry["out"], entry["loss"]

    def _sched_entry(self, base, dataset):
        """The graph of variant `base` that draws its own batches from `dataset` (captured by train_step_resident), or None."""
        entry = self._graphs.get(base + ("sched",))
        return entry if entry is not None and entry["sched"].dataset is dataset else None

    def train_step_resident(self, dataset):
        """One iteration on a device-resident dataset (anything with next_batch() -> (indices, model_input, ground_truth) and
        write_batch(dst_input, dst_gt)): once the whole-iteration graph of the variant exists, the batch is gathered straight into
        its static input block -- no intermediate batch tensors, no copies -- and the graph is replayed.
        A dataset that also offers scheduled_draw() / peek_batch() (datasets/pixel_sampler.py: DeviceSchedule) has the draw captured as the graph's
        FIRST NODE (graphs keyed (..., "sched")): an iteration is then one graph launch and nothing else on the stream -- a launch between two
        replays left the chip idle for ~14 us around it."""
        if self.use_graph and hasattr(dataset, "write_batch") and self._full_graph_ok():
            base = ("full", self.model.wants_background(self.iter_step), self.iter_step >= self.add_objectvio_iter)
            sched_ok = self.draw_in_graph and hasattr(dataset, "scheduled_draw") and hasattr(dataset, "peek_batch")
            entry = self._sched_entry(base, dataset) if sched_ok else None
            if entry is not None:
                sd = entry["sched"]
                self.model.train()
                if self._draw_ahead:
                    sd.before_replay_ahead()
                    entry["graph"].replay()
                    sd.after_replay_ahead()
                else:
                    sd.before_replay()
                    entry["graph"].replay()
                    sd.after_replay()
            else:
                entry = self._graphs.get(base)
                if entry is not None:
                    self.model.train()
                    dataset.write_batch(entry["static"]["input"], entry["static"]["gt"])
                    entry["graph"].replay()
                elif sched_ok and base + ("sched",) not in self._graphs:
                    self._capture_dataset = dataset         # the capture takes the draw in; its first replay draws the batch peeked at here
                    try:
                        return self.train_step(*dataset.peek_batch())
                    finally:
                        self._capture_dataset = None
            if entry is not None:
                self.model.ray_sampler._rounds = entry["rounds"]
                self._after_replay()
                self.iter_step += 1
                return entry["out"], entry["loss"]
        return self.train_step(*dataset.next_batch())

    def _capture(self, key, fresh):
        """fresh: dict of live tensors with the shapes of this variant; becomes the static input block."""
        with_bg, call_reg = key
        st = {"rays": {k: v.clone() for k, v in fresh["rays"].items()}, "z_vals": fresh["z_vals"].clone(), "z_eik": fresh["z_eik"].clone(),
              "gt": {k: v.clone() for k, v in fresh["gt"].items()}}
        if with_bg:
            st["bg"] = {k: v.clone() for k, v in fresh["bg"].items()}
        self._warm_up(lambda: self._graph_body(st, with_bg, call_reg))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=self._capture_mode()):
            out, loss_out = self._graph_body(st, with_bg, call_reg)
        entry = {"graph": g, "static": st, "out": out, "loss": loss_out}
        self._graphs[key] = entry
        return entry

    @staticmethod
    def _copy_into(dst, src):
        for k, v in src.items():
            dst[k].copy_(v, non_blocking=True)

    def _train_step_graph(self, indices, model_input, ground_truth):
        if self._full_graph_ok():
            return self._train_step_full_graph(model_input, ground_truth)
        model = self.model
        model.train()
        with torch.no_grad():
            rays = model.prepare_rays(model_input)
            z_vals, z_eik = model.sample(rays)
            with_bg = model.wants_background(self.iter_step)
            bg = model.prepare_background(model_input) if with_bg else None
        key = (with_bg, self.iter_step >= self.add_objectvio_iter)
        fresh = {"rays": rays, "z_vals": z_vals, "z_eik": z_eik, "gt": {k: v for k, v in ground_truth.items()}, "bg": bg}
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._capture(key, fresh)   # capture replays nothing: run the body once more via replay below
        st = entry["static"]
        # all live inputs -> the graph's static input block in ONE multi-tensor copy (was ~14 copy launches in the
        # launch-bound stretch between the sampler's last host sync and the graph launch)
        pairs = [(st["z_vals"], z_vals), (st["z_eik"], z_eik)]
        pairs += [(st["rays"][k], rays[k]) for k in ("ray_dirs", "cam_loc", "depth_scale", "rot")]
        pairs += [(st["gt"][k], v) for k, v in fresh["gt"].items()]
        if with_bg:
            pairs += [(st["bg"][k], v) for k, v in bg.items() if torch.is_tensor(v)]
        groups = {}
        for d, s_ in pairs:
            groups.setdefault((d.dtype, s_.dtype), ([], []))
            groups[(d.dtype, s_.dtype)][0].append(d)
            groups[(d.dtype, s_.dtype)][1].append(s_)
        for dl, sl in groups.values():
            torch._foreach_copy_(dl, sl)
        entry["graph"].replay()
        self._after_replay()
        self.iter_step += 1
        return entry["out"], entry["loss"]
