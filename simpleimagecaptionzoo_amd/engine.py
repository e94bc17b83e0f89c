"""Engine subclasses for the hot path (drop-ins for ModelEngines/BUTD_Engine.py on top of Engine.py).

Mirrors the reference's Engine methods that sit on the path -- same names, argument meaning, batch-tuple layouts
and output contract:
    modify_visual_inputs          BUTD_Engine.py:23-47
    training_epoch                Engine.py:169-188      (XE: forward, label-smoothed loss, backward, clamp 0.1, Adam)
    SCST_training_epoch           Engine.py:251-272      (greedy baseline, sampled rollout, CIDEr-D reward, REINFORCE,
                                                          clamp 0.25, Adam)
    eval_captions_json_generation Engine.py:274-300      (greedy or beam decode -> [{'image_id', 'caption'}])
Everything between the batch tuple and the updated parameters runs in libicz; the host only moves the feature batch
to the device and (for evaluation) turns ids into words.  With torch.distributed initialised (dist.py) the same
methods run data-parallel: per-rank shards, all-reduced loss normaliser and gradients.
"""
import json

import numpy as np
import torch

from ._lib import BUTD_PARAM_KEYS, check, lib, ptr, stream_ptr
from .captioner import BUTDDetection_Captioner
from .ciderd import CiderDReward
from . import dist as icz_dist
from .features import wait_event


class FusedAdam:
    """clip_gradient (Utils.py:241-250) + torch.optim.Adam(betas=(0.9,0.999), eps=1e-8, weight_decay=0)
    (Utils.py:219-220) as one HIP kernel per parameter tensor.  Exposes param_groups like a torch optimizer so the
    reference's set_lr / get_lr helpers (Utils.py:231-239) keep working."""

    def __init__(self, params, lr):
        if isinstance(params, (list, tuple)) and params and isinstance(params[0], dict):
            self.param_groups = [dict(g) for g in params]
            for g in self.param_groups:
                g.setdefault("lr", lr)
        else:
            self.param_groups = [{"params": list(params), "lr": lr}]
        self.state = {}

    def zero_grad(self):
        pass

    def state_dict(self):
        """Same layout as torch.optim.Adam.state_dict(): parameters are numbered in param_groups order.  (The reference
        neither saves the optimizer, Engine.py:81-88, nor keeps it across epochs, Engine.py:133-136; with this a caller
        can -- SURVEY.md 8f row 4.)"""
        index, groups = {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                index.setdefault(p, len(index))
                ids.append(index[p])
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "betas": (0.9, 0.999), "eps": 1e-8,
                           "weight_decay": 0, "params": ids})
        state = {index[p]: {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"].clone(),
                            "exp_avg_sq": st["exp_avg_sq"].clone()} for p, st in self.state.items()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        params = [p for g in self.param_groups for p in g["params"]]
        if [len(g["params"]) for g in sd["param_groups"]] != [len(g["params"]) for g in self.param_groups]:
            raise ValueError("optimizer state has a different parameter grouping")
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g["lr"] = saved["lr"]
        self.state = {}
        for i, st in sd["state"].items():
            p = params[int(i)]
            self.state[p] = {"step": int(float(st["step"])), "exp_avg": st["exp_avg"].to(p.device, torch.float32).clone(),
                             "exp_avg_sq": st["exp_avg_sq"].to(p.device, torch.float32).clone()}

    def step_with(self, grads_by_param, clip):
        """grads_by_param: {parameter: gradient tensor}.  One multi-tensor launch per param group."""
        import ctypes as C
        for group in self.param_groups:
            ps = [p for p in group["params"] if grads_by_param.get(p) is not None]
            if not ps:
                continue
            steps = set()
            for p in ps:
                st = self.state.get(p)
                if st is None:
                    st = {"step": 0, "exp_avg": torch.zeros_like(p.data), "exp_avg_sq": torch.zeros_like(p.data)}
                    self.state[p] = st
                st["step"] += 1
                steps.add(st["step"])
            assert len(steps) == 1, "parameters of one group share the step count"
            n = len(ps)
            for lo in range(0, n, 32):
                chunk = ps[lo:lo + 32]
                m = len(chunk)
                arr = lambda xs: (C.c_void_p * m)(*xs)
                check(lib().icz_adam_clamp_multi(
                    m, arr([p.data.data_ptr() for p in chunk]), arr([grads_by_param[p].data_ptr() for p in chunk]),
                    arr([self.state[p]["exp_avg"].data_ptr() for p in chunk]),
                    arr([self.state[p]["exp_avg_sq"].data_ptr() for p in chunk]),
                    (C.c_int64 * m)(*[p.numel() for p in chunk]), float(group["lr"]), float(clip), steps.pop() if lo + 32 >= n else next(iter(steps)),
                    stream_ptr()))


def init_optimizer(optimizer_type, params, learning_rate):
    """Utils.py:222-229 (Adam only on the fused path)."""
    if optimizer_type != "Adam":
        raise ValueError("the fused optimiser implements Adam (the reference's default, Main.py:171)")
    if len(params) == 0:
        return None
    return FusedAdam(params, learning_rate)


class Engine(object):
    """The part of Engine.py:16-41 the hot path needs (construction, device, vocabulary, tag)."""

    def __init__(self, model_settings_json, dataset_name, caption_vocab, data_dir=None, use_bu="unused", device="cuda:0",
                 cider_df=None, max_batch=128):
        if isinstance(model_settings_json, dict):
            self.settings = dict(model_settings_json)
        else:
            self.settings = json.load(open(model_settings_json, "r"))
        self.device = torch.device(device)
        self.data_dir = data_dir
        self.dataset_name = dataset_name
        self.use_bu = use_bu
        self.caption_vocab = caption_vocab
        self.tag = "Model_" + self.settings["model_type"] + "_Dataset_" + dataset_name
        self.model = self.model_construction(max_batch)
        self.model.to(self.device)
        self.cnn_ft_model = 0
        self._cider_df = cider_df
        self._scorer = None
        self._pinned = None
        self._dev_feats = None
        self.use_graphs = True      # replay captured hipGraphs in SCST / greedy evaluation (buffers are persistent)
        # data-parallel: start each gradient group's all-reduce from the library's gradient-ready callback, beside the rest of the backward
        # pass (default); False (or ICZ_DP_OVERLAP=0) = ONE all-reduce of the flat buffer behind the backward pass -- the A/B leg
        # bench.py reports as `dp_overlap`
        import os
        self.dp_overlap = os.environ.get("ICZ_DP_OVERLAP", "1") not in ("0", "")
        self.phase_events = None    # a list: every SCST step appends its phase-boundary events (phase_times() turns them into ms)
        # The hot path runs on its own non-default stream: after hipGraph replays, eager launches on the legacy null
        # stream were measured 2-3x slower on ROCm 7.2 (implicit synchronisation with the graph's internal streams).
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def model_construction(self, max_batch):
        raise NotImplementedError

    # ---- checkpoints: the reference's files (Engine.py:43-70, 81-88) + the optimizer state next to them
    def _cp_paths(self, scst, root=None):
        import os
        flag = "scst_" if scst else ""
        cp_dir = os.path.join(root or "./CheckPoints/%s/" % self.tag, "cp")
        return cp_dir, os.path.join(cp_dir, "Captioner_%scp.pth" % flag), os.path.join(cp_dir, "%sstate_histories.json" % flag), \
            os.path.join(cp_dir, "Optimizer_%scp.pth" % flag)

    def save_checkpoint(self, cider_scores, save_scst_model=False, optimizer=None, root=None):
        """Engine.py:81-88: model state_dict + score history, same file names; with `optimizer` also its state
        (`Optimizer_[scst_]cp.pth`, torch.optim layout) -- the reference re-creates Adam every epoch (Engine.py:133-136) and
        loses the moments at every restart."""
        import os
        cp_dir, model_path, his_path, opt_path = self._cp_paths(save_scst_model, root)
        os.makedirs(cp_dir, exist_ok=True)
        torch.save(self.model.state_dict(), model_path)
        json.dump({"cider_his": cider_scores}, open(his_path, "w"))
        if optimizer is not None:
            torch.save(optimizer.state_dict(), opt_path)

    def load_from_checkpoint(self, load_scst_model=False, load_best=False, optimizer=None, root=None):
        """Engine.py:43-70 (same files, same return value) + the optimizer state when `optimizer` is given and the file exists."""
        import os
        cp_dir, model_path, his_path, opt_path = self._cp_paths(load_scst_model, root)
        cider_his, start_epoch = [], 1
        best_not_found = False
        if load_best:
            best = os.path.join(os.path.dirname(cp_dir), "best", os.path.basename(model_path))
            if os.path.exists(best):
                self.model.load_state_dict(torch.load(best, map_location=self.device))
            else:
                best_not_found = True
        if not load_best or best_not_found:
            if os.path.exists(his_path):
                cider_his = json.load(open(his_path, "r"))["cider_his"]
            if os.path.exists(model_path):
                self.model.load_state_dict(torch.load(model_path, map_location=self.device))
            else:
                print("recent checkpoint not found.")
            if optimizer is not None and os.path.exists(opt_path):
                optimizer.load_state_dict(torch.load(opt_path, map_location=self.device))
            start_epoch = len(cider_his) + 1
        return cider_his, start_epoch


class BUTDDetection_Eng(Engine):
    """ModelEngines/BUTD_Engine.py:21-47 + the three hot Engine methods on libicz."""

    def model_construction(self, max_batch):
        s = self.settings
        assert s["model_type"] in ("BUTDDetection", "BUTDSpatial")
        return BUTDDetection_Captioner(atten_dim=s["atten_dim"], embed_dim=s["embed_dim"], hidden_dim=s["hidden_dim"],
                                       vocab_size=len(self.caption_vocab), device=str(self.device),
                                       num_regions=s.get("num_regions", 36), enc_dim=s.get("enc_dim", 2048),
                                       max_batch=max_batch)

    # ---- E4 -------------------------------------------------------------------------------------------------
    def modify_visual_inputs(self, img_tensors, supp_info_datas=None):
        """BUTD_Engine.py:23-47: stack per-image (n_i, D) features into (B, max_n, D) fp32 + mask (None if all
        rows are full).  Staged through a reusable pinned buffer and copied asynchronously."""
        if isinstance(supp_info_datas, dict) and torch.is_tensor(supp_info_datas.get("bu_feats")):
            # extension: a batch already resident in HBM (prefetching loaders, bench.py); padded batches carry their counts
            feats, counts = supp_info_datas["bu_feats"], supp_info_datas.get("bu_counts")
            if self.use_graphs and counts is None and feats.is_cuda:
                # captured graphs are keyed by the feature tensor's address: a prefetching loader's ring of `depth` buffers replays
                # them, any other tensor is copied (device to device, ~7 us for 19 MB) into ONE persistent buffer first.  The
                # address set belongs to one loader instance: a new one (e.g. per epoch) starts it afresh
                ring = supp_info_datas.get("bu_ring")
                token, depth = ring if isinstance(ring, tuple) else (None, 0)
                cache = self.__dict__.setdefault("_feat_ring", {"token": None, "addrs": set()})
                if token is not None and cache["token"] != token:
                    cache["token"], cache["addrs"] = token, set()
                seen = cache["addrs"] if token is not None else ()
                if feats.data_ptr() not in seen:
                    if token is not None and len(seen) < depth:
                        seen.add(feats.data_ptr())
                    else:
                        buf = getattr(self, "_dev_batch", None)
                        if buf is None or buf.shape != feats.shape:
                            buf = self._dev_batch = torch.empty_like(feats, memory_format=torch.contiguous_format)
                        buf.copy_(feats, non_blocking=True)
                        feats = buf
            out = {"bu_feats": feats, "bu_bboxes": supp_info_datas.get("bu_bboxes"), "bu_masks": None}
            if counts is not None:
                out["bu_masks"] = (torch.arange(feats.shape[1]).unsqueeze(0) < torch.tensor(counts).unsqueeze(1)).float().to(self.device)
                out["bu_counts"] = list(counts)
            return out
        bu_feats = [s["bu_feat"] for s in supp_info_datas]
        bu_bboxes = [s["bu_bbox"] for s in supp_info_datas]
        counts = [int(f.shape[0]) for f in bu_feats]
        max_len = max(counts)
        B, D = len(bu_feats), bu_feats[0].shape[1]
        # flat staging buffers sized for the largest batch seen: 'adaptive' batches change max_len every step
        need = B * max_len * D
        if self._pinned is None or self._pinned.numel() < need:
            self._pinned = torch.zeros(need, dtype=torch.float32).pin_memory()
            self._dev_feats = torch.empty(need, dtype=torch.float32, device=self.device)
        if getattr(self, "_h2d_done", None) is not None:
            wait_event(self._h2d_done)           # the previous batch's copy has left the pinned buffer
        host = self._pinned[:need].view(B, max_len, D)
        hv = host.numpy()
        for i, f in enumerate(bu_feats):
            hv[i, :counts[i]] = f
            if counts[i] < max_len:
                hv[i, counts[i]:] = 0
        # a persistent device buffer: a stable address lets the captured graphs be replayed
        feats = self._dev_feats[:need].view(B, max_len, D)
        feats.copy_(host, non_blocking=True)
        self._h2d_done = torch.cuda.Event()
        self._h2d_done.record()
        if min(counts) == max_len:
            return {"bu_feats": feats, "bu_bboxes": bu_bboxes, "bu_masks": None}
        masks = (torch.arange(max_len).unsqueeze(0) < torch.tensor(counts).unsqueeze(1)).float().to(self.device)
        # `bu_counts` (extension): the counts behind the prefix mask, so that the Captioner need not read the mask back
        return {"bu_feats": feats, "bu_bboxes": bu_bboxes, "bu_masks": masks, "bu_counts": counts}

    # ---- helpers ------------------------------------------------------------------------------------------
    # Gradient groups in the order the backward pass completes them (include/icz.h: icz_butd_set_grad_callback); the flat
    # buffer is laid out group by group so that each group is one contiguous slice = one all-reduce.
    _GRAD_STAGES = (("predict.weight_v", "predict.weight_g", "predict.bias"),
                    ("embed.0.weight", "TD_atten.weight_ih", "TD_atten.weight_hh"),
                    ("language_model.weight_ih", "language_model.weight_hh"))

    def _grads(self):
        """Gradient buffers as views into ONE flat fp32 buffer (few large all-reduces over xGMI move them all)."""
        if getattr(self, "_flat", None) is None:
            named = self._trainable()
            staged = [k for st in self._GRAD_STAGES for k in st if k in named]
            order = staged + [k for k in named if k not in staged]
            offs, total, bounds = {}, 0, []
            for k in order:
                offs[k] = total
                total += (named[k].numel() + 63) // 64 * 64
                bounds.append(total)
            self._flat = torch.zeros(total, dtype=torch.float32, device=self.device)
            self._gviews = {k: self._flat[offs[k]:offs[k] + named[k].numel()].view_as(named[k]) for k in named}
            # slice of the flat buffer per stage (+ the remainder), only meaningful when every staged key is present
            self._stage_slices = []
            if len(staged) == sum(len(st) for st in self._GRAD_STAGES):
                lo, i = 0, 0
                for st in self._GRAD_STAGES:
                    i += len(st)
                    self._stage_slices.append((lo, bounds[i - 1]))
                    lo = bounds[i - 1]
                self._stage_slices.append((lo, total))
        return self._gviews

    def _reduce_grads_begin(self, h):
        """Data-parallel: start the all-reduce of each gradient group as soon as the backward pass has enqueued it, so that
        RCCL moves it over xGMI beside the remaining weight-gradient GEMMs (the hook is a no-op for one process)."""
        self._pending = []
        if not icz_dist.is_distributed() or not self._stage_slices or not hasattr(h, "set_grad_callback"):
            return False
        if not self.dp_overlap:
            if getattr(self, "_hooked", None) is h:
                h.set_grad_callback(None)
                self._hooked = None
            return False
        if getattr(self, "_hooked", None) is not h:
            import torch.distributed as td

            def on_ready(stage):
                try:      # called from inside the C library: an exception cannot propagate through it
                    lo, hi = self._stage_slices[stage]
                    self._pending.append(td.all_reduce(self._flat[lo:hi], op=td.ReduceOp.SUM, async_op=True))
                except Exception as e:      # re-raised by _reduce_grads_end
                    self._hook_error = e
            h.set_grad_callback(on_ready)
            self._hooked = h
        return True

    def _reduce_grads_end(self, overlapped):
        if not icz_dist.is_distributed():
            return
        if not overlapped:
            icz_dist.all_reduce_sum_(self._flat)
            return
        err, self._hook_error = getattr(self, "_hook_error", None), None
        if err is not None or len(self._pending) != len(self._stage_slices) - 1:
            raise RuntimeError("gradient hook: %d of %d groups were reduced%s" % (
                len(self._pending), len(self._stage_slices) - 1, "" if err is None else " (%r)" % (err,)))
        lo, hi = self._stage_slices[-1]
        icz_dist.all_reduce_sum_(self._flat[lo:hi])
        for w in self._pending:
            w.wait()
        self._pending = []

    def _features(self, visual_inputs):
        """The device tensor the decoder handle consumes (bottom-up features here; NIC: the image embedding)."""
        return visual_inputs["bu_feats"]

    def _trainable(self):
        """name -> parameter for everything the optimizer updates (AoA: the decoder only, AoA_Model.py:669-674)."""
        return getattr(self.model, "_trainable", self.model._named)()

    def _apply(self, optimizer, clip):
        named = self._trainable()
        grads = self._gviews
        if isinstance(optimizer, FusedAdam):
            optimizer.step_with({named[k]: grads[k] for k in grads}, clip)
        else:   # a torch optimizer handed in by unmodified reference code
            for k in grads:
                named[k].grad = grads[k].clamp(-clip, clip)
            optimizer.step()

    def scorer(self):
        if self._scorer is None:
            df = self._cider_df
            if df is None:
                raise RuntimeError("SCST needs the CIDEr document-frequency table: pass cider_df={'document_frequency':"
                                   " ..., 'ref_len': n} (the content of cider/data/<dataset>-train.p)")
            if isinstance(df, str):
                import pickle
                df = pickle.load(open(df, "rb"), encoding="latin1")
            self._scorer = CiderDReward(df["document_frequency"], df["ref_len"], self.caption_vocab.word2ix, self.device)
            self._scorer.persistent = True
        return self._scorer

    def _hot_handle(self):
        h = self.model._handle()
        if self.use_graphs != h._persistent:
            h.enable_graphs(self.use_graphs)
        return h

    # ---- E1 -------------------------------------------------------------------------------------------------
    def training_epoch(self, dataloader, optimizer, criterion, tqdm_visible=True, rngs=None):
        with _on_stream(self):
            return self._training_epoch(dataloader, optimizer, criterion, tqdm_visible, rngs)

    def _training_epoch(self, dataloader, optimizer, criterion, tqdm_visible=True, rngs=None):
        """Engine.py:169-188.  `criterion` is the reference's LabelSmoothingLoss (only its .smoothing is read: the
        loss and its gradient are fused into the backward kernels).  `rngs` (tests) supplies one icz_rng per batch."""
        self.model.train()
        smoothing = float(getattr(criterion, "smoothing", 0.0))
        monitor = _monitor(dataloader, "Training Process", tqdm_visible)
        losses = []
        for batch_i, (img_ids, img_tensors, captions, lengths, supp_info_datas) in enumerate(monitor):
            visual_inputs = self.modify_visual_inputs(img_tensors, supp_info_datas)
            lengths = [cap_len - 1 for cap_len in lengths]
            h = self.model._handle()
            rng = rngs[batch_i] if rngs is not None else self.model._next_rng()
            h.xe_forward(self._features(visual_inputs), captions, lengths, rng, train=True)
            grads = self._grads()
            n_glob = 0.0
            if icz_dist.is_distributed():
                # G2: every rank scales by 1 / (global token count).  The count is all-reduced on the device
                # and handed to the library as a device scalar: no host round trip between forward and backward
                n_dev = torch.full((1,), float(sum(lengths)), dtype=torch.float32, device=self.device)
                if hasattr(h, "set_mask_sum_global"):
                    icz_dist.all_reduce_sum_(n_dev)
                    h.set_mask_sum_global(n_dev)
                    n_glob = -1.0
                else:
                    n_glob = icz_dist.all_reduce_scalar(n_dev)
            ov = self._reduce_grads_begin(h)
            loss = h.xe_backward(grads, smoothing, n_glob)
            self._reduce_grads_end(ov)
            self._apply(optimizer, 0.1)
            losses.append(loss)
            if tqdm_visible:
                monitor.set_postfix(Loss=np.round(loss.item(), decimals=4))
        return losses

    # ---- E2 -------------------------------------------------------------------------------------------------
    def SCST_training_epoch(self, dataloader, optimizer, criterion, tqdm_visible=True, rngs=None):
        with _on_stream(self):
            return self._scst_training_epoch(dataloader, optimizer, criterion, tqdm_visible, rngs)

    def _scst_training_epoch(self, dataloader, optimizer, criterion, tqdm_visible=True, rngs=None):
        """Engine.py:251-272: greedy baseline (eval mode) + multinomial rollout (train mode) + CIDEr-D reward +
        REINFORCE + clamp 0.25 + Adam, all on the device; `criterion` (RewardCriterion) is implied."""
        self.model.train()
        scorer = self.scorer()
        # References of images the scorer has not seen yet (the whole first epoch) are cooked on a loader thread one batch ahead
        # of the step that needs them (Utils.py:319-367 cooks them inside the step, for every batch of every epoch); a loader
        # that is not a prefetcher already is wrapped in one, which also stages host-side features through pinned buffers
        dataloader, restore = _cook_ahead(dataloader, scorer, self.device, getattr(self, "cook_ahead", True),
                                          self.__dict__.setdefault("_cook_cache", {}))
        monitor = _monitor(dataloader, "Training Process", tqdm_visible)
        losses = []
        try:
            self._scst_steps(monitor, scorer, optimizer, rngs, tqdm_visible, losses)
        finally:
            restore()
        return losses

    def _mark(self, marks):
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)

    def phase_times(self, skip=0):
        """Mean GPU time per phase (ms) of the SCST steps recorded since `phase_events = []`: rollouts / reward / backward /
        allreduce_exposed (what the step WAITS for the gradient exchange behind its backward pass: 0 for one process, the whole
        all-reduce with dp_overlap off) / adam.  Call after a synchronize."""
        names = ("rollouts", "reward", "backward", "allreduce_exposed", "adam")
        steps = (self.phase_events or [])[skip:]
        if not steps:
            return {}
        return {n: sum(m[i].elapsed_time(m[i + 1]) for m in steps) / len(steps) for i, n in enumerate(names)}

    def _scst_steps(self, monitor, scorer, optimizer, rngs, tqdm_visible, losses):
        for batch_i, (img_ids, img_tensors, img_gts, supp_info_datas) in enumerate(monitor):
            visual_inputs = self.modify_visual_inputs(img_tensors=img_tensors, supp_info_datas=supp_info_datas)
            feats = self._features(visual_inputs)
            h = self._hot_handle()
            rng = rngs[batch_i] if rngs is not None else self.model._next_rng()
            marks = [] if self.phase_events is not None else None
            self._mark(marks)
            greedy_res, seq_gen, seq_logprobs = h.rollouts(feats, 20, rng)
            self._mark(marks)
            rewards = scorer.reward(seq_gen, greedy_res, img_gts, img_ids)
            self._mark(marks)
            grads = self._grads()
            msum_glob = 0.0
            if icz_dist.is_distributed():
                ms = h.sample_mask_sum()
                if torch.is_tensor(ms) and hasattr(h, "set_mask_sum_global"):
                    # all-reduced on the device and handed over as a device scalar: no host round trip before backward
                    icz_dist.all_reduce_sum_(ms)
                    h.set_mask_sum_global(ms)
                    msum_glob = -1.0
                else:
                    msum_glob = icz_dist.all_reduce_scalar(ms)
            ov = self._reduce_grads_begin(h)
            loss, _ = h.sample_backward(rewards, grads, msum_glob)
            self._mark(marks)
            self._reduce_grads_end(ov)
            self._mark(marks)
            self._apply(optimizer, 0.25)
            self._mark(marks)
            if marks is not None:
                self.phase_events.append(marks)
            losses.append(loss.clone())      # with graphs the handle returns one persistent buffer, overwritten by the next step
            if tqdm_visible:
                monitor.set_postfix(Loss=np.round(loss.item(), decimals=4))

    # ---- E3 -------------------------------------------------------------------------------------------------
    def eval_captions_json_generation(self, dataloader, eval_beam_size=-1, tqdm_visible=True):
        with _on_stream(self):
            return self._eval_captions_json_generation(dataloader, eval_beam_size, tqdm_visible)

    def _eval_captions_json_generation(self, dataloader, eval_beam_size=-1, tqdm_visible=True):
        """Engine.py:274-300.  Beam search accepts any batch size here (the reference's loader uses 1).
        Data-parallel (torch.distributed initialised): rank r decodes the batches i with i % world == r (and
        loads only those where the loader allows, _rank_batches); the (image id, token ids) rows are all-gathered and every rank returns the
        complete list in loader order -- what the corpus-level scorer after it (COCO_Eval_Utils.py:15-35) needs in one
        place; rank 0 is the one that should write / score it."""
        self.model.eval()
        print("Generating captions json for evaluation. Beam Search: %s" % (eval_beam_size != -1))
        dp = icz_dist.is_distributed()
        rank, world = icz_dist.rank(), icz_dist.world_size()
        # data-parallel: this rank's batches only.  An indexable loader (list, Dataset-like: __len__ + __getitem__) or one with a
        # shard(rank, world) method is never asked for the other ranks' batches (no feature I/O for them); any other iterable is
        # walked in full and the foreign batches dropped
        monitor = _monitor(_rank_batches(dataloader, rank, world) if dp else enumerate(dataloader), "Generating Process", tqdm_visible)
        ids_out, rows_out, keys_out = [], [], []
        for batch_i, (image_ids, img_tensors, supp_info_datas) in monitor:
            nb = len(image_ids)
            visual_inputs = self.modify_visual_inputs(img_tensors=img_tensors, supp_info_datas=supp_info_datas)
            h = self._hot_handle()
            if eval_beam_size != -1:
                seqs, lens = h.beam_search(self._features(visual_inputs), eval_beam_size, 50)
                seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
                rows = [seqs[i, :lens[i]] for i in range(len(lens))]
            else:
                rows = list(h.greedy(self._features(visual_inputs), 20).cpu().numpy())
            ids_out += [int(i) for i in image_ids]
            rows_out += rows
            keys_out += [(batch_i << 20) + j for j in range(nb)]      # loader order: batch index, then row
        if dp:
            ids_out, rows_out = icz_dist.gather_caption_rows(keys_out, ids_out, rows_out, self.device)
        result = []
        ix2word = self.caption_vocab.ix2word
        for image_id, sampled_ids in zip(ids_out, rows_out):
            sampled_caption = []
            for word_id in sampled_ids:
                word = ix2word[int(word_id)]
                if word == "<end>":
                    break
                elif word != "<sta>":
                    sampled_caption.append(word)
            result.append({"image_id": image_id, "caption": " ".join(sampled_caption)})
        return result


class AoADetection_Eng(BUTDDetection_Eng):
    """ModelEngines/AoA_Engine.py (same visual-input handling as the BUTD engine) + the three hot Engine methods.
    `use_bu='adaptive'` (10..100 boxes per image, Main.py:158): the handle is sized for 100 regions and every batch carries
    its region counts (AoA_Engine.py:37-46 builds the equivalent prefix masks)."""

    # the AoA library's callback stages (include/icz.h: icz_aoa_set_grad_callback): 41 MB + 92 MB of the 163 MB of decoder gradients are
    # on the wire before the backward call returns; the attention block and h_norm (30 MB) follow after it
    _GRAD_STAGES = (("decoder.predict.weight_v", "decoder.predict.weight_g", "decoder.predict.bias"),
                    ("decoder.embed.0.weight", "decoder.lstm.weight_ih", "decoder.lstm.weight_hh", "decoder.lstm.bias_ih", "decoder.lstm.bias_hh"))

    def model_construction(self, max_batch):
        from .aoa import AoADetection_Captioner
        s = self.settings
        assert s["model_type"] in ("AoADetection", "AoASpatial")
        regions = s.get("num_regions", 100 if self.use_bu == "adaptive" else 36)
        return AoADetection_Captioner(vocab_size=len(self.caption_vocab), num_heads=s.get("num_heads", 8), hidden_dim=s["hidden_dim"],
                                      embed_dim=s["embed_dim"], device=str(self.device), num_regions=regions,
                                      enc_dim=s.get("enc_dim", 2048), max_batch=max_batch)

    def _features(self, visual_inputs):
        return self.model._feats(visual_inputs)


class NIC_Eng(BUTDDetection_Eng):
    """ModelEngines/NIC_Engine.py (= the base Engine) with the three hot methods on the NIC decoder handle.  The CNN encoder +
    img_embedding of NIC_Model.py:8-37 is outside the path: batches carry the image embedding -- `supp_info_datas =
    {'img_feats': (B, embed_dim) tensor}` -- or the Captioner was given an `encoder` module for `img_tensors`."""

    def model_construction(self, max_batch):
        from .nic import NICDecoder_Captioner
        s = self.settings
        assert s["model_type"] == "NIC"
        return NICDecoder_Captioner(embed_dim=s["embed_dim"], hidden_dim=s["hidden_dim"], vocab_size=len(self.caption_vocab),
                                    device=str(self.device), max_batch=max_batch)

    def modify_visual_inputs(self, img_tensors, supp_info_datas=None):
        if isinstance(supp_info_datas, dict) and torch.is_tensor(supp_info_datas.get("img_feats")):
            return {"img_feats": supp_info_datas["img_feats"].to(self.device, torch.float32)}
        return {"img_tensors": img_tensors.to(self.device)}            # Engine.py:32-41

    def _features(self, visual_inputs):
        return self.model._features(visual_inputs).detach().contiguous()


class BUTDSpatial_Eng(BUTDDetection_Eng):
    """Same decoder over a 7x7x2048 grid fed as precomputed features (49 regions); the CNN encoder of
    BUTD_Model.py:8-38 is outside the hot path."""

    def model_construction(self, max_batch):
        self.settings.setdefault("num_regions", self.settings.get("enc_img_size", 7) ** 2)
        return super().model_construction(max_batch)


class _on_stream:
    """Run a block on the engine's stream, ordered after / before the caller's current stream."""

    def __init__(self, eng):
        self.eng = eng

    def __enter__(self):
        self.outer = torch.cuda.current_stream(self.eng.device)
        self.eng.stream.wait_stream(self.outer)
        self.ctx = torch.cuda.stream(self.eng.stream)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        self.outer.wait_stream(self.eng.stream)
        return False


def _rank_batches(loader, rank, world):
    """(batch index, batch) of the batches i with i % world == rank, loading as little else as the loader allows."""
    if hasattr(loader, "shard"):
        for i, b in loader.shard(rank, world):
            yield i, b
    elif hasattr(loader, "__getitem__") and hasattr(loader, "__len__"):
        for i in range(rank, len(loader), world):
            yield i, loader[i]
    else:
        for i, b in enumerate(loader):
            if i % world == rank:
                yield i, b


def _cook_ahead(loader, scorer, device, enabled, cache=None):
    """-> (loader whose worker thread cooks unseen references one batch ahead, restore()).  `cache` (a dict the Engine keeps) holds
    the wrapping prefetcher across epochs: its pinned ring, copy stream and thread pool are set up once."""
    from .features import DevicePrefetcher
    cook = lambda batch: scorer.prepare(batch[0], batch[2])
    if not enabled:
        return loader, lambda: None
    if isinstance(loader, DevicePrefetcher):
        if loader.on_batch is not None:
            return loader, lambda: None
        loader.on_batch = cook

        def restore():
            loader.on_batch = None
        return loader, restore
    pf = cache.get("pf") if cache is not None else None
    if pf is None:
        pf = DevicePrefetcher(loader, device, on_batch=cook)
        if cache is not None:
            cache["pf"] = pf
    pf.loader, pf.on_batch = loader, cook

    def release():
        pf.loader = None            # do not keep the caller's loader alive between epochs
    return pf, release


def _monitor(dataloader, desc, visible):
    if not visible:
        return dataloader
    import tqdm
    return tqdm.tqdm(dataloader, desc=desc)
