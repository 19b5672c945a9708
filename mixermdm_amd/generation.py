"""Callers on either side of the denoising loop, mirroring the reference's harnesses:

  * ``generate_one_sample`` = LitGenModel.generate_loop + the save layout of generate_one_sample
    (src/scripts/infer/mixermdm.py:57-144): model(batch) -> output[0] as [T, 2, 262] -> gaussian_filter1d(sigma=1, time axis,
    mode="nearest") -> ``<name>_motion.npy`` / ``_influence{1,2}.npy``;
  * ``generate_for_evaluation`` = the generation loop of EvaluationDataset* (src/evaluation/datasets.py:71-163): per item a
    ``forward_test`` on B = 1 (or mm_num_repeats) motions of that item's length, reshape to [B, T, 2, 262], optional
    de-normalisation, zero padding to ``max_length``.

Smoothing runs on the GPU (``ops.gaussian_filter1d``, bit-compatible with scipy's double-accumulating correlate1d);
file writing and list bookkeeping stay on the host, as in the reference.  Plots / viewers are out of scope.
"""
import copy
import os
import numpy as np
import torch

from . import ops


def generate_loop(model, batch, window_size):
    """LitGenModel.generate_loop (infer/mixermdm.py:102-144).  `batch` may carry 'cond'/'x_T' instead of prompts."""
    batch = copy.copy(batch)
    batch["motion_lens"] = torch.full((1, 1), int(window_size), dtype=torch.long)
    for src, dst in (("prompt_individual1", "text_individual1"), ("prompt_individual2", "text_individual2"), ("prompt_interaction", "text_interaction")):
        if src in batch:
            batch[dst] = [batch[src]]
    out = model(batch)
    motion = out["output"][0].reshape(out["output"][0].shape[0], 2, -1)          # [T, 2, 262]
    motion = ops.gaussian_filter1d(motion.reshape(1, motion.shape[0], -1), 1.0).reshape(motion.shape)
    return (motion.cpu().numpy(), out["influence_i1"], out["influence_i2"], out["out1"], out["out2"], out["out_influenced"])


def generate_one_sample(model, batch, name, save_folder, window_size=299):
    """infer/mixermdm.py:57-99 without the plotting: writes <name>_motion.npy, _influence1.npy, _influence2.npy."""
    motion, i1, i2, _, _, _ = generate_loop(model, batch, window_size)
    os.makedirs(save_folder, exist_ok=True)
    path = os.path.join(save_folder, name)
    np.save(path + "_motion.npy", motion)
    np.save(path + "_influence1.npy", np.array([t.cpu().numpy() for t in i1]))
    np.save(path + "_influence2.npy", np.array([t.cpu().numpy() for t in i2]))
    return motion


def generate_for_evaluation(model, items, max_length=300, mm_idxs=(), mm_num_repeats=1, normalizer=None, extended=True, shard_items=False,
                            batching="sequential", inflight=2, max_rows=4800, max_items=64, seed=None):
    """Generation loop of the evaluation datasets (datasets.py:71-163).

    batching: "sequential" = the reference's loop, one ``forward_test`` per item; "inflight" / "ragged" = the same calls handed to
    ``model.sample_many`` (several items in flight over one weight set / the items packed into ragged batches with per-sequence lengths on the
    device) -- every motion is bit-identical to the sequential loop's on the same x_T; measured (bench.py --eval-items, fp32): "ragged" 1.9 x the
    sequential loop (3.0 x in fp32_split), "inflight" 1.00-1.03 x (kept for completeness, not a throughput lever).  "sequential" streams the
    items (a lazy iterable or DataLoader is consumed one item at a time, as the reference's loop does); the other two need the calls at hand
    together and materialise the batches first.
    seed: when given and an item carries no 'x_T', its noise is drawn from ``Generator(seed + item index)`` -- the same motions whatever the
    batching, the sharding or the world size (without it every call draws from the device's global generator, as the reference does).

    shard_items=True under an initialised torch.distributed process group (one process per GPU): the items are dealt round-robin to the
    ranks, every rank generates its own (no collective during sampling), and the per-item results are merged in item order on every
    rank -- the 8 GPUs of a node do one evaluation pass instead of 8 identical ones.  x_T must then come from the items (or differ by
    rank seed): each rank draws its own noise, as independent evaluation items do in the reference -- or pass `seed`: item i's noise is
    then a function of (seed, i) alone and sharded and unsharded passes produce the same motions.

    items: iterable of dicts with 'text' (tuple/list of str), 'motion_lens' (LongTensor [1]), optionally
    'text_individual1/2', and -- since the CLIP tower is upstream -- optionally a precomputed 'cond' [1, 8*768].
    Returns (generated_motions, mm_generated_motions) with the reference's dictionary keys.
    """
    generated, mm_generated = [], []
    mm_idxs = set(mm_idxs)
    if shard_items:
        from . import distributed as D
        items = list(items)
        mine = set(D.shard_items(len(items)))
        gen_l, mm_l = {}, {}
    def calls():
        for i, data in enumerate(items):
            if shard_items and i not in mine:
                continue
            rep = mm_num_repeats if i in mm_idxs else 1
            batch = {"text": list(data["text"]) * rep, "motion_lens": data["motion_lens"]}
            if extended:
                batch["text_individual1"] = list(data["text_individual1"]) * rep
                batch["text_individual2"] = list(data["text_individual2"]) * rep
            if "cond" in data:
                batch["cond"] = data["cond"].repeat(rep, 1)
            if "x_T" in data:
                batch["x_T"] = data["x_T"]
            elif seed is not None:
                T = int(data["motion_lens"][0])
                gen = torch.Generator().manual_seed(int(seed) + i)
                batch["x_T"] = torch.randn(rep, T, 524, generator=gen)
            yield i, data, batch
    with torch.no_grad():
        if batching != "sequential":
            todo = list(calls())
            outs = model.sample_many([b for _, _, b in todo], mode="eval_intermediate", batching=batching, inflight=inflight, max_rows=max_rows,
                                     max_items=max_items, keep_history=False)
            stream = ((i, data, r["output"]) for (i, data, _), r in zip(todo, outs))
        else:
            stream = ((i, data, model.forward_test(batch)["output"]) for i, data, batch in calls())
        for i, data, out in stream:
            motions = out.reshape(out.shape[0], out.shape[1], 2, -1).cpu().numpy()
            if normalizer is not None:
                motions = normalizer.backward(motions)
            B, T = motions.shape[:2]
            if T < max_length:
                motions = np.concatenate((motions, np.zeros((B, max_length - T, 2, motions.shape[-1]))), axis=1)
            assert motions.shape[1] == max_length
            sub = {"motion1": motions[0, :, 0], "motion2": motions[0, :, 1], "motion_lens": data["motion_lens"][0], "text": data["text"][0]}
            if extended:
                sub.update(text_individual1=data["text_individual1"][0], text_individual2=data["text_individual2"][0])
            if shard_items:
                gen_l[i] = sub
            else:
                generated.append(sub)
            if i in mm_idxs:
                mm = {"mm_motions": motions, "motion_lens": data["motion_lens"][0], "text": data["text"][0]}
                if extended:
                    mm.update(text_individual1=data["text_individual1"][0], text_individual2=data["text_individual2"][0])
                if shard_items:
                    mm_l[i] = mm
                else:
                    mm_generated.append(mm)
    if shard_items:
        dev = getattr(model, "device", None)
        generated = D.gather_items(gen_l, len(items), device=dev)
        order = sorted(j for j in mm_idxs if j < len(items))
        merged = D.gather_items({order.index(j): v for j, v in mm_l.items()}, len(order), device=dev)
        mm_generated = merged
    return generated, mm_generated
