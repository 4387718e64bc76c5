"""End-to-end parity of the host modules + HIP kernels against goldens produced by the reference
(teacher pseudo-labelling, student step, EMA, tracker post-processing).  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from oracle import spot_oracle as O
from spot_amd.detinit import det_fill_, det_input
from spot_amd.synthetic import Settings, sup_batch, unsup_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def G(a):
    return T(a).to(DEV)


@pytest.fixture(params=["fp32", "bf16x3-linear-fwd", "bf16x3"])
def linear_fwd(request):
    """The arithmetic modes of the GEMM products against the same goldens at the same tolerances: the exact fp32
    product path and the bf16x3 experiment (operands cut in three bf16 pieces, six piece products) for the forward
    nn.Linear products only and for every GEMM launched from Python (forward and both gradients)."""
    from spot_amd import ops
    prev, ops.MATMUL_PRECISION = ops.MATMUL_PRECISION, request.param
    yield request.param
    ops.MATMUL_PRECISION = prev


def build(seed, dropout=0.1):
    from spot_amd.transt import transt_resnet50
    net = transt_resnet50(Settings(device="cpu", dropout=dropout))
    det_fill_(net, seed)
    return net.to(DEV)


def test_teacher_pseudo_label_pipeline_golden(golden):
    from spot_amd.actor import TranstSparseActor
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    g = golden("g5_teacher")
    B, seed = int(g["B"]), int(g["seed"])
    S = Settings()
    net = build(seed).eval()
    with torch.no_grad():
        net.iou_embed.layers[2].bias += float(g["iou_bias_shift"])
    actor = TranstSparseActor(S, net, net, None)
    data = TensorDict({
        "key_images": G(det_input("key_images", (B, 3, 128, 128), seed, 0, 256, np.uint8)),
        "unlabeled_images": G(det_input("unlabeled_images", (B, 3, 256, 256), seed, 0, 256, np.uint8)),
        "key_avg": G(det_input("key_avg", (B, 3), seed, 0, 255)),
        "unlabeled_avg": G(det_input("unlabeled_avg", (B, 3), seed, 0, 255)),
        "key_anno": G(det_input("key_anno", (B, 4), seed, 20, 80)),
        "unlabeled_anno": G(det_input("unlabeled_anno", (B, 4), seed, 40, 120))})
    with torch.no_grad():
        raw = net(teacher_data=TensorDict(data))
    # score maps / boxes: 1e-3 relative (fp32), as the parity contract states
    np.testing.assert_allclose(raw["pred_logits"].cpu().numpy(), g["pred_logits"], rtol=1e-3, atol=5e-4)
    np.testing.assert_allclose(raw["pred_boxes"].cpu().numpy(), g["pred_boxes"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(raw["pred_ious"].cpu().numpy(), g["pred_ious"], rtol=1e-3, atol=1e-4)
    pseudo = actor(mode="inference", teacher_data=TensorDict(data))
    np.testing.assert_allclose(pseudo["best_bbox"].cpu().numpy(), g["best_bbox"], atol=0.5)       # 0.5 px
    np.testing.assert_allclose(pseudo["best_score"].cpu().numpy(), g["best_score"], rtol=1e-3)
    np.testing.assert_allclose(pseudo["best_conf"].cpu().numpy(), g["best_conf"], rtol=1e-3)
    xy, wh = det_input("unl_xy", (B, 1, 2), seed, 0, 400), det_input("unl_wh", (B, 1, 2), seed, 20, 200)
    ue = np.concatenate([xy, wh], -1).astype(np.float32)
    xy, wh = det_input("aug_xy", (B, 1, 2), seed, 0, 400), det_input("aug_wh", (B, 1, 2), seed, 20, 200)
    ae = np.concatenate([xy, wh], -1).astype(np.float32)
    mid = {"unlabeled_box_extract": T(ue), "unlabeled_resize_factors": T(det_input("unl_rf", (B,), seed, 0.5, 2.0)),
           "aug_box_extract": T(ae), "aug_resize_factors": T(det_input("aug_rf", (B,), seed, 0.5, 2.0))}
    res = SparseMiddleware(S).record_relay_results(mid, pseudo)
    assert np.array_equal(res["aug_success"].cpu().numpy(), g["aug_success"])                     # identical keep mask
    np.testing.assert_allclose(res["aug_anno"].cpu().numpy(), g["aug_anno"], atol=0.5)


def test_teacher_pseudo_label_pipeline_golden_at_benchmark_batch(golden, linear_fwd):
    """The benchmark configuration's 32-clip pseudo-labelling pass against the reference's outputs: every 4th token of
    the maps, every decoded box / score / confidence, the re-projected boxes and the keep mask."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    from tests.test_oracle_model import teacher_b32_inputs
    g = golden("g5_teacher_b32")
    B, seed, st = int(g["B"]), int(g["seed"]), int(g["slice_step"])
    S = Settings()
    net = build(seed).eval()
    with torch.no_grad():
        net.iou_embed.layers[2].bias += float(g["iou_bias_shift"])
    actor = TranstSparseActor(S, net, net, None)
    key, unl, mid = teacher_b32_inputs(g)
    data = TensorDict({"key_images": key.to(DEV), "unlabeled_images": unl.to(DEV),
                       "key_avg": G(det_input("key_avg", (B, 3), seed, 0, 255)),
                       "unlabeled_avg": G(det_input("unlabeled_avg", (B, 3), seed, 0, 255)),
                       "key_anno": G(det_input("key_anno", (B, 4), seed, 20, 80)),
                       "unlabeled_anno": G(det_input("unlabeled_anno", (B, 4), seed, 40, 120))})
    with torch.no_grad():
        raw = net(teacher_data=TensorDict(data))
    np.testing.assert_allclose(raw["pred_logits"].cpu().numpy()[:, ::st], g["pred_logits"], rtol=1e-3, atol=5e-4)
    np.testing.assert_allclose(raw["pred_boxes"].cpu().numpy()[:, ::st], g["pred_boxes"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(raw["pred_ious"].cpu().numpy()[:, ::st], g["pred_ious"], rtol=1e-3, atol=1e-4)
    pseudo = actor(mode="inference", teacher_data=TensorDict(data))
    np.testing.assert_allclose(pseudo["best_bbox"].cpu().numpy(), g["best_bbox"], atol=0.5)       # 0.5 px
    np.testing.assert_allclose(pseudo["best_score"].cpu().numpy(), g["best_score"], rtol=1e-3)
    np.testing.assert_allclose(pseudo["best_conf"].cpu().numpy(), g["best_conf"], rtol=1e-3)
    res = SparseMiddleware(S).record_relay_results(mid, pseudo)
    assert np.array_equal(res["aug_success"].cpu().numpy(), g["aug_success"])                     # identical keep mask
    assert 0 < int(g["aug_success"].sum()) < B
    np.testing.assert_allclose(res["aug_anno"].cpu().numpy(), g["aug_anno"], atol=0.5)


def test_teacher_pseudo_label_pipeline_train_mode_golden_at_benchmark_batch(golden):
    """The pass the benchmark times: the teacher left in the module-default train() mode, as the reference's trainer
    leaves it (base_sparse_actor.py:42-49 toggles the student only) — batch-statistics BatchNorm in both backbone
    passes, whose running statistics move — at 32 clips with dropout 0, against the reference run the same way:
    maps, decoded boxes / scores / confidences, identical keep mask, and the BatchNorm buffers after the pass (the
    template pass runs on the second stream with its buffer updates deferred: they must land as in the sequential
    order)."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    from tests.test_oracle_model import check_batchnorm_buffers, teacher_b32_inputs
    g = golden("g23_teacher_train_b32")
    B, seed, st = int(g["B"]), int(g["seed"]), int(g["slice_step"])
    S = Settings(dropout=0.0)
    net = build(seed, dropout=0.0).train()
    with torch.no_grad():
        net.iou_embed.layers[2].bias += float(g["iou_bias_shift"])
    start = {k: v.clone() for k, v in net.state_dict().items()}
    actor = TranstSparseActor(S, net, net, None)
    key, unl, mid = teacher_b32_inputs(g)
    data = TensorDict({"key_images": key.to(DEV), "unlabeled_images": unl.to(DEV),
                       "key_avg": G(det_input("key_avg", (B, 3), seed, 0, 255)),
                       "unlabeled_avg": G(det_input("unlabeled_avg", (B, 3), seed, 0, 255)),
                       "key_anno": G(det_input("key_anno", (B, 4), seed, 20, 80)),
                       "unlabeled_anno": G(det_input("unlabeled_anno", (B, 4), seed, 40, 120))})
    with torch.no_grad():
        raw = net(teacher_data=TensorDict(data))
    assert net.training
    np.testing.assert_allclose(raw["pred_logits"].cpu().numpy()[:, ::st], g["pred_logits"], rtol=1e-3, atol=5e-4)
    np.testing.assert_allclose(raw["pred_boxes"].cpu().numpy()[:, ::st], g["pred_boxes"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(raw["pred_ious"].cpu().numpy()[:, ::st], g["pred_ious"], rtol=1e-3, atol=1e-4)
    check_batchnorm_buffers(g, net.state_dict())
    net.load_state_dict(start)                      # the actor's pass below starts from the same buffers
    pseudo = actor(mode="inference", teacher_data=TensorDict(data))
    np.testing.assert_allclose(pseudo["best_bbox"].cpu().numpy(), g["best_bbox"], atol=0.5)       # 0.5 px
    np.testing.assert_allclose(pseudo["best_score"].cpu().numpy(), g["best_score"], rtol=1e-3)
    np.testing.assert_allclose(pseudo["best_conf"].cpu().numpy(), g["best_conf"], rtol=1e-3)
    res = SparseMiddleware(S).record_relay_results(mid, pseudo)
    assert np.array_equal(res["aug_success"].cpu().numpy(), g["aug_success"])                     # identical keep mask
    assert 0 < int(g["aug_success"].sum()) < B
    np.testing.assert_allclose(res["aug_anno"].cpu().numpy(), g["aug_anno"], atol=0.5)


def test_burnin_step_golden(golden):
    """actor(mode="burnin") (transt_sparse_actor.py:100-138) against the reference: loss, every statistic, every
    gradient norm and the gradient samples of the fusion network and the heads; then the trainer's burn-in cycle
    (ltr_sparse_trainer.py:49-88,255-275) takes the same step: masks all False, backward, optimizer step."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.criterion import transt_loss
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    from spot_amd.trainer import LTRSparseTrainer
    from tests.test_oracle_model import burnin_batch, check_gradient_samples
    g = golden("g24_burnin")
    seed, bs = int(g["seed"]), int(g["bs"])
    S = Settings(dropout=0.0, batch_size_sup=bs, batch_size=bs)
    net = build(seed, dropout=0.0).train()
    actor = TranstSparseActor(S, net, net, transt_loss(S))
    batch = lambda: TensorDict({k: v.to(DEV) for k, v in burnin_batch(seed, bs).items()})
    data = batch()
    data["masks"] = torch.zeros(bs, dtype=torch.bool, device=DEV)
    loss, stats = actor(mode="burnin", burnin_data=data)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-3)
    stats = actor.stats_to_float(stats)
    ref_stats = dict(zip([str(k) for k in g["stat_keys"]], g["stat_vals"]))
    assert set(stats) == set(ref_stats)
    for k, v in ref_stats.items():
        np.testing.assert_allclose(stats[k], v, rtol=2e-3, atol=1e-5, err_msg=k)
    loss.backward()
    gn = dict(zip([str(k) for k in g["gn_keys"]], g["gn_vals"]))
    params = dict(net.named_parameters())
    worst = max(abs(float(p.grad.norm()) - gn[k]) / gn[k] for k, p in params.items() if gn.get(k, 0) > 1e-6)
    assert worst < 2e-2, worst
    check_gradient_samples(g, {k: p.grad for k, p in params.items() if p.grad is not None})
    # the trainer's burn-in cycle on the same batch: one optimizer step along the golden's gradient
    net2 = build(seed, dropout=0.0).train()
    before = {k: p.detach().clone() for k, p in net2.named_parameters()}
    actor2 = TranstSparseActor(S, net2, net2, transt_loss(S))
    opt = torch.optim.SGD(net2.parameters(), lr=1e-3)
    trainer = LTRSparseTrainer(actor2, None, opt, S, SparseMiddleware(S))
    out = trainer.cycle_dataset_burnin([batch()])
    np.testing.assert_allclose(float(out["Loss/total"]), float(g["loss"]), rtol=1e-3)
    assert trainer.num_frames == bs and trainer.skipped == 0
    for k in ("class_embed.layers.2.weight", "featurefusion_network.decoder.layers.0.linear2.weight"):
        step = (before[k] - dict(net2.named_parameters())[k].detach()) / 1e-3
        ref = params[k].grad
        cos = float((step * ref).sum() / (step.norm() * ref.norm()))
        assert cos > 0.9999, (k, cos)


def test_sharded_pseudo_labelling_equals_single_batch():
    """Pseudo-labelling + filtering of a clip set cut into contiguous per-rank blocks (dist.shard_range) gives the
    result of one big batch: the teacher runs in eval mode, so clips do not interact and no collective is needed."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.dist import shard_range
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    S = Settings()
    net = build(21).eval()
    actor = TranstSparseActor(S, net, net, None)
    mw = SparseMiddleware(S)
    n_clips, world = 48, 4
    data, mid = unsup_batch(n_clips, 9, device=DEV)

    def run(lo, hi):
        d = TensorDict({k: v[lo:hi] for k, v in data.items() if k.startswith(("key_", "unlabeled_"))})
        m = {k: v[lo:hi] for k, v in mid.items()}
        pseudo = actor(mode="inference", teacher_data=d)
        return pseudo, mw.record_relay_results(m, pseudo)

    with torch.no_grad():
        # centre the confidences on the threshold so both outcomes occur
        conf = run(0, n_clips)[0]["best_conf"].clamp(1e-6, 1 - 1e-6)
        net.iou_embed.layers[2].bias += float(np.log(0.85 / 0.15)) - torch.log(conf / (1 - conf)).median()
        full_p, full_r = run(0, n_clips)
        parts = [run(*shard_range(n_clips, r, world)) for r in range(world)]
    # no kernel on this path lets clips interact or picks its arithmetic by batch size: bit-identical results
    assert 0 < int(full_r["aug_success"].sum()) < n_clips
    for k in ("best_bbox", "best_score", "best_conf"):
        assert torch.equal(torch.cat([p[k] for p, _ in parts]), full_p[k]), k
    for k in ("aug_success", "aug_anno"):
        assert torch.equal(torch.cat([r[k] for _, r in parts]), full_r[k]), k


def test_student_step_golden(golden, linear_fwd):
    from spot_amd.actor import TranstSparseActor
    from spot_amd.criterion import transt_loss
    from spot_amd.tensordict import TensorDict
    from tests.test_oracle_model import student_batches
    g = golden("g15_student")
    seed, bs, bu = int(g["seed"]), int(g["bs"]), int(g["bu"])
    S = Settings(dropout=0.0)
    net = build(seed, dropout=0.0).train()
    actor = TranstSparseActor(S, net, net, transt_loss(S))
    sup, sp = student_batches(seed, bs, bu)
    sup = TensorDict({k: v.to(DEV) for k, v in sup.items()})
    sp = TensorDict({k: v.to(DEV) for k, v in sp.items()})
    loss, stats = actor(mode="spsup", burnin_data=sup, sp_data=sp)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-3)
    stats = actor.stats_to_float(stats)
    ref_stats = dict(zip([str(k) for k in g["stat_keys"]], g["stat_vals"]))
    for k, v in ref_stats.items():
        np.testing.assert_allclose(stats[k], v, rtol=2e-3, atol=1e-5, err_msg=k)
    loss.backward()
    gn = dict(zip([str(k) for k in g["gn_keys"]], g["gn_vals"]))
    params = dict(net.named_parameters())
    worst = max(abs(float(p.grad.norm()) - gn[k]) / gn[k] for k, p in params.items() if gn.get(k, 0) > 1e-6)
    assert worst < 2e-2, worst
    # 64-element slices.  With every GEMM of the backward pass in bf16x3 pieces the slice of the stem's weight gradient
    # (the end of the longest chain of train-mode BatchNorm backward passes) has 2 of 64 elements 2.6e-4 away from the
    # reference where the exact path stays within 1.1e-4, at a tensor maximum of 5.5e-2: a different rounding pattern
    # amplified by that chain, not a larger error per product (per GEMM the pieces are at or below the exact kernel's
    # error against float64, tools/probes/bf16x3_error.py).  That mode gets 3x the absolute tolerance on these slices
    # only; every other check of this test is shared.
    slack = 3.0 if linear_fwd == "bf16x3" else 1.0
    for k in [n[len("gslice."):] for n in g.files if n.startswith("gslice.")]:
        ref = g["gslice." + k]
        np.testing.assert_allclose(params[k].grad.reshape(-1)[:64].cpu().numpy(), ref, rtol=2e-2,
                                   atol=slack * 2e-3 * max(1e-6, float(np.abs(ref).max())), err_msg=k)
    # every gradient tensor of the fusion network, input_proj and the heads against the reference's samples of it:
    # cosine >= 0.9999 each, largest deviation <= 2e-3 of the tensor's largest sample
    from tests.test_oracle_model import check_gradient_samples
    check_gradient_samples(g, {k: p.grad for k, p in params.items() if p.grad is not None})


def test_two_stream_forward_matches_sequential_including_running_stats():
    """Template pass on a second stream with deferred BatchNorm running-stat updates == the sequential order."""
    from spot_amd import fusion
    from spot_amd.tensordict import TensorDict
    from tests.test_oracle_model import student_batches
    outs = []
    saved = fusion.TWO_STREAMS
    try:
        for two in (False, True):
            fusion.TWO_STREAMS = two
            net = build(13, dropout=0.0).train()
            sup, sp = student_batches(13, 2, 2)
            sup = TensorDict({k: v.to(DEV) for k, v in sup.items()})
            sp = TensorDict({k: v.to(DEV) for k, v in sp.items()})
            with torch.no_grad():
                for _ in range(2):      # two iterations: the second starts from updated running buffers
                    out = net(burnin_data=TensorDict(sup), sp_data=TensorDict(sp))
            torch.cuda.synchronize()
            outs.append((out, {k: v.detach().clone() for k, v in net.state_dict().items() if "running" in k or
                               "tracked" in k}))
    finally:
        fusion.TWO_STREAMS = saved
    (o0, s0), (o1, s1) = outs
    for k in s0:
        if "tracked" in k:
            assert torch.equal(s0[k], s1[k]), k
        else:       # batch statistics are summed by float atomics whose order differs from run to run
            np.testing.assert_allclose(s0[k].cpu().numpy(), s1[k].cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    assert int(s1["backbone.0.body.bn1.num_batches_tracked"]) == 4      # search + template, twice
    for k in ("logits_sup", "boxes_unsup"):
        np.testing.assert_allclose(o0[k].cpu().numpy(), o1[k].cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_student_step_bf16_attention_tracks_fp32(golden):
    """Training configuration with bf16 attention products (fp32 softmax/accumulate, fp32 elsewhere): loss within
    1e-2 relative of the reference's fp32 loss, gradients within cosine 0.999 of the fp32 run's."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.criterion import transt_loss
    from spot_amd.tensordict import TensorDict
    from tests.test_oracle_model import student_batches
    g = golden("g15_student")
    seed, bs, bu = int(g["seed"]), int(g["bs"]), int(g["bu"])
    S = Settings(dropout=0.0)
    grads, losses = [], []
    for dtype in ("f32", "bf16"):
        net = build(seed, dropout=0.0).train()
        net.featurefusion_network.set_attention_dtype(dtype)
        actor = TranstSparseActor(S, net, net, transt_loss(S))
        sup, sp = student_batches(seed, bs, bu)
        sup = TensorDict({k: v.to(DEV) for k, v in sup.items()})
        sp = TensorDict({k: v.to(DEV) for k, v in sp.items()})
        loss, _ = actor(mode="spsup", burnin_data=sup, sp_data=sp)
        loss.backward()
        losses.append(loss.item())
        grads.append(torch.cat([p.grad.flatten() for _, p in net.named_parameters() if p.grad is not None]))
    np.testing.assert_allclose(losses[1], float(g["loss"]), rtol=1e-2)
    cos = float((grads[0].double() @ grads[1].double()) / (grads[0].double().norm() * grads[1].double().norm()))
    assert cos >= 0.999, cos
    assert not torch.equal(grads[0], grads[1])          # the bf16 path really ran


def test_gradient_sinks_match_autograd_accumulation(golden):
    """Gradients added into the flat buffer by the kernels equal the ones autograd accumulates, and a
    second backward adds on top (the trainer zeroes the buffer once per step)."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.arena import ParamArena
    from spot_amd.criterion import transt_loss
    from spot_amd.tensordict import TensorDict
    from tests.test_oracle_model import student_batches
    S = Settings(dropout=0.0)
    grads = []
    for use_arena in (False, True):
        net = build(11, dropout=0.0).train()
        ready = []
        if use_arena:
            arena = ParamArena(net, with_grads=True)
            arena.add_ready_listener(lambda p: ready.append(id(p)))
        actor = TranstSparseActor(S, net, net, transt_loss(S))
        for rep in range(2 if use_arena else 1):
            sup, sp = student_batches(11, 1, 2)
            sup = TensorDict({k: v.to(DEV) for k, v in sup.items()})
            sp = TensorDict({k: v.to(DEV) for k, v in sp.items()})
            loss, _ = actor(mode="spsup", burnin_data=sup, sp_data=sp)
            loss.backward()
            if rep == 0:
                grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        if use_arena:
            twice = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            assert len(set(ready)) > 100          # fusion + head parameters reported completion
    plain, sunk = grads
    assert plain.keys() <= sunk.keys()
    bb_a, bb_b = [], []
    for k, gp in plain.items():
        scale = float(gp.abs().max()) + 1e-12
        if "backbone" in k:
            # these sit behind the whole fusion backward, whose atomic sums are order-dependent, and behind
            # BatchNorm over a 3-clip batch: run-to-run spread alone is a few percent of the largest element
            bb_a.append(gp.flatten())
            bb_b.append(sunk[k].flatten())
            continue
        assert float((gp - sunk[k]).abs().max()) <= 2e-3 * scale + 1e-7, k
        assert float((twice[k] - 2 * sunk[k]).abs().max()) <= 2e-2 * scale + 1e-7, k
    a, b = torch.cat(bb_a).double(), torch.cat(bb_b).double()
    assert float((a @ b) / (a.norm() * b.norm())) > 0.999


def test_ema_over_arena_matches_oracle():
    from spot_amd.actor import TranstSparseActor
    stu, tea = build(5), build(6)
    tea.backbone[0].body.bn1.num_batches_tracked.fill_(100)
    stu.backbone[0].body.bn1.num_batches_tracked.fill_(7)
    before_t = {k: v.detach().cpu().clone() for k, v in tea.state_dict().items()}
    before_s = {k: v.detach().cpu().clone() for k, v in stu.state_dict().items()}
    actor = TranstSparseActor(Settings(), stu, tea, None)
    actor(mode="update", keep_rate=0.998)
    ref = O.ema_update(before_t, before_s, 0.998)
    for k, v in tea.state_dict().items():
        assert torch.equal(v.cpu(), ref[k]), k
    for k, v in stu.state_dict().items():
        assert torch.equal(v.cpu(), before_s[k]), k
    actor(mode="update", keep_rate=0)
    for k, v in tea.state_dict().items():
        assert torch.equal(v.cpu().float(), before_s[k].float()), k


def test_trainer_iteration_end_to_end():
    """Teacher -> decode -> filter -> compaction -> student step runs on device and updates the student;
    compaction matches the oracle rule on the kernel's own keep mask."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.criterion import transt_loss
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.trainer import LTRSparseTrainer
    S = Settings(dropout=0.0, batch_size_sup=2, batch_size_unsup=6, teacher_update_frequency=1)
    S.curriculum_settings["confidence_threshold"] = 0.5
    stu, tea = build(7, 0.0), build(7, 0.0)
    actor = TranstSparseActor(S, stu, tea, transt_loss(S))
    opt = torch.optim.AdamW([p for p in stu.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    trainer = LTRSparseTrainer(actor, None, opt, S, SparseMiddleware(S))
    data_unsup, mid = unsup_batch(6, 3, device=DEV)
    data_sup = sup_batch(2, 3, device=DEV)
    pseudo = trainer._pseudo_label_generation(data_unsup)
    aug = trainer._middleware_ops(mid, pseudo)
    _, n_keep, idx = trainer.compact(data_unsup, aug)
    keep = aug["aug_success"].cpu().numpy()
    ref_idx = O.compact_indices(keep, 6)
    if ref_idx is None:
        assert int(n_keep.item()) == 0
    else:
        assert np.array_equal(idx.cpu().numpy(), ref_idx.astype(np.int32))
    w0 = stu.class_embed.layers[2].weight.detach().clone()
    stats = trainer.train_iteration(0, data_sup, data_unsup, mid)
    if keep.any():
        assert stats is not None and torch.isfinite(stats["Loss/total"])
        assert not torch.equal(w0, stu.class_embed.layers[2].weight.detach())


def test_tracker_post_processing_golden(golden):
    from spot_amd.tracker import TransT_Tracker
    g = golden("g11_tracker_post")

    class Net:
        def initialize(self):
            pass

        def template(self, z):
            pass

        def track(self, x):
            return {"pred_logits": G(g["logits"][self.i]), "pred_boxes": G(g["boxes"][self.i])}

    net = Net()
    trk = TransT_Tracker("g", net, window_penalty=0.49, exemplar_size=128, instance_size=256)
    img = np.zeros((int(g["image_hw"][0]), int(g["image_hw"][1]), 3), np.uint8)
    trk.initialize(img, {"init_bbox": [float(v) for v in g["init_bbox"]]})
    for i in range(g["logits"].shape[0]):
        net.i = i
        o = trk.track(img)
        np.testing.assert_allclose(o["target_bbox"], g["target_bbox"][i], rtol=1e-6, atol=1e-3)
        np.testing.assert_allclose(np.max(o["best_score"]), g["best"][i], rtol=1e-6)
        np.testing.assert_allclose(trk.center_pos, g["center"][i], rtol=1e-6, atol=1e-3)
        np.testing.assert_allclose(trk.size, g["size"][i], rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("use_graph", [False, True])
def test_online_template_track_golden(golden, use_graph):
    """TransT.template / TransT.track at the network boundary against the reference's (transt_model.py:86-117 inside
    the tracker of transt_tracker.py:108-193): the network holder's frozen-weight path (kept position tables and
    transformed weights, folded inference BatchNorm, split-key attention), eagerly and, from the third frame, as the
    replayed HIP graph.  The crops stand in for get_subwindow's result on both sides."""
    from spot_amd.tracker import NetWithBackbone, TransT_Tracker
    g = golden("g21_track")
    seed, n = int(g["seed"]), int(g["n_frames"])

    class Recording(NetWithBackbone):
        def track(self, x):
            self.__dict__["last"] = NetWithBackbone.track(self, x)       # inside a graph: the replayed output buffers
            return self.__dict__["last"]

    net = Recording(net=build(seed), use_gpu=True)
    trk = TransT_Tracker("g", net, window_penalty=0.49, exemplar_size=128, instance_size=256, use_graph=use_graph)
    crops = [G(det_input("z_crop", (1, 3, 128, 128), seed, 0, 256, np.uint8))]
    crops += list(G(det_input("x_crops", (n, 3, 256, 256), seed, 0, 256, np.uint8)).unsqueeze(1))
    trk.get_subwindow = lambda im, pos, msz, osz, avg: crops.pop(0)
    img = np.zeros((int(g["image_hw"][0]), int(g["image_hw"][1]), 3), np.uint8)
    trk.initialize(img, {"init_bbox": [float(v) for v in g["init_bbox"]]})
    for i in range(n):
        o = trk.track(img)
        last = net.__dict__["last"]
        np.testing.assert_allclose(last["pred_logits"].cpu().numpy()[0], g["pred_logits"][i], rtol=1e-3, atol=5e-4)
        np.testing.assert_allclose(last["pred_boxes"].cpu().numpy()[0], g["pred_boxes"][i], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(o["target_bbox"], g["target_bbox"][i], atol=0.5)                # 0.5 px
        np.testing.assert_allclose(np.max(o["best_score"]), g["best"][i], rtol=1e-3)
        np.testing.assert_allclose(trk.center_pos, g["center"][i], atol=0.5)
        np.testing.assert_allclose(trk.size, g["size"][i], atol=0.5)
    assert (trk._graph is not None) == use_graph


def test_tracker_graph_replay_equals_eager():
    """From the third frame the device pipeline replays as a captured graph: same boxes and score maps."""
    from spot_amd.tracker import NetWithBackbone, TransT_Tracker
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 255, size=(7, 256, 256, 3), dtype=np.uint8)
    for y in range(7):
        frames[y, 90 + 3 * y:150 + 3 * y, 100 + 2 * y:160 + 2 * y] = 235
    runs = []
    for use_graph in (False, True):
        trk = TransT_Tracker("t", NetWithBackbone(net=build(1), use_gpu=True), 0.49, 128, 256, use_graph=use_graph)
        trk.initialize(frames[0], {"init_bbox": [100.0, 90.0, 60.0, 60.0]})
        runs.append([trk.track(f) for f in frames[1:]])
        if use_graph:
            assert trk._graph is not None
            trk.initialize(frames[0], {"init_bbox": [100.0, 90.0, 60.0, 60.0]})     # new sequence: graph dropped
            assert trk._graph is None
            again = [trk.track(f) for f in frames[1:]]
            np.testing.assert_allclose(again[-1]["target_bbox"], runs[-1][-1]["target_bbox"], rtol=1e-4, atol=1e-3)
    for a, b in zip(*runs):
        np.testing.assert_allclose(a["target_bbox"], b["target_bbox"], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(a["best_score"], b["best_score"], rtol=1e-3, atol=1e-5)


def test_tracker_runs_on_synthetic_sequence():
    """cfg-1 shape: 256x256 frames, template 128^2 cached once, track() per frame."""
    from spot_amd.tracker import NetWithBackbone, TransT_Tracker
    net = NetWithBackbone(net=build(1), use_gpu=True)
    trk = TransT_Tracker("t", net, 0.49, 128, 256)
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 255, size=(4, 256, 256, 3), dtype=np.uint8)
    trk.initialize(frames[0], {"init_bbox": [96.0, 96.0, 64.0, 64.0]})
    for f in frames[1:]:
        o = trk.track(f)
        assert len(o["target_bbox"]) == 4 and np.all(np.isfinite(o["target_bbox"]))
        assert o["best_score"].shape == (1024,)


def test_student_forward_with_in_network_augmentation():
    """The reference's training configuration (grey-scale, rotation, flips): the student forward applies the
    transforms to the pseudo-labelled pairs only, hands the transformed boxes on as targets, and trains."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.criterion import transt_loss
    from spot_amd.detinit import det_fill_
    from spot_amd.synthetic import Settings
    from spot_amd.tensordict import TensorDict
    from spot_amd.transt import transt_resnet50
    from tests.test_oracle_model import student_batches
    S = Settings(device="cpu", dropout=0.0, strong_augmentation_tag=True)
    S.aug_settings.update(gray=0.5, angle=10, lr_f_prob=0.5, ud_f_prob=0.5)
    assert transt_resnet50(S, network="teacher").independent_trans is None
    net = det_fill_(transt_resnet50(S), 23).cuda().train()
    assert net.independent_trans is not None and net.joint_trans is not None
    sup, sp = student_batches(23, 2, 6)
    sup = TensorDict({k: v.cuda() for k, v in sup.items()})
    sp = TensorDict({k: v.cuda() for k, v in sp.items()})
    sup["masks"] = torch.zeros(2, dtype=torch.bool, device="cuda")
    sp["masks"] = torch.ones(6, dtype=torch.bool, device="cuda")
    anno_in, sup_anno_in = sp["aug_anno"].clone(), sup["search_anno"].clone()
    torch.manual_seed(5)
    Sg = Settings(dropout=0.0)
    actor = TranstSparseActor(Sg, net, net, transt_loss(Sg))
    loss, stats = actor(mode="spsup", burnin_data=TensorDict(sup), sp_data=TensorDict(sp))
    assert torch.isfinite(loss)
    loss.backward()
    out = net(burnin_data=TensorDict(sup), sp_data=TensorDict(sp))
    assert torch.equal(out["search_anno"], sup_anno_in)                   # labelled pairs: boxes untouched
    assert not torch.equal(out["aug_anno"], anno_in)                      # pseudo-labelled pairs: boxes follow
    assert (out["aug_anno"] >= 0).all()



def test_all_failed_batch_is_skipped_and_leaves_the_student_untouched():
    """No clip passes the confidence threshold: the iteration is skipped as the reference does before its student forward
    (ltr_sparse_trainer.py:328-332): parameters, BatchNorm running statistics and counters stay bit-identical, no
    optimizer step happens, and the next good iteration trains normally."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.criterion import transt_loss
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.trainer import LTRSparseTrainer
    S = Settings(dropout=0.0, batch_size_sup=2, batch_size_unsup=4, teacher_update_frequency=1000)
    S.curriculum_settings["confidence_threshold"] = 2.0            # a sigmoid output never reaches it
    stu, tea = build(7, 0.0), build(7, 0.0)
    actor = TranstSparseActor(S, stu, tea, transt_loss(S))
    opt = torch.optim.AdamW([p for p in stu.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    trainer = LTRSparseTrainer(actor, None, opt, S, SparseMiddleware(S))
    data_unsup, mid = unsup_batch(4, 3, device=DEV)
    data_sup = sup_batch(2, 3, device=DEV)
    before = {k: v.detach().clone() for k, v in stu.state_dict().items()}
    assert trainer.train_iteration(0, data_sup, data_unsup, mid) is None and trainer.skipped == 1
    for k, v in stu.state_dict().items():
        assert torch.equal(v, before[k]), k
    trainer.middleware.confidence_threshold = 0.0
    S.curriculum_settings["confidence_threshold"] = 0.0
    trainer.middleware = SparseMiddleware(S)
    stats = trainer.train_iteration(1, sup_batch(2, 3, device=DEV), data_unsup, mid)
    assert stats is not None and torch.isfinite(stats["Loss/total"])
    sd = stu.state_dict()
    assert not torch.equal(sd["class_embed.layers.2.weight"], before["class_embed.layers.2.weight"])
    assert int(sd["backbone.0.body.bn1.num_batches_tracked"]) == int(before["backbone.0.body.bn1.num_batches_tracked"]) + 2


def test_clip_grad_norm_matches_torch():
    from spot_amd import ops
    g = torch.randn(1_000_448, device=DEV) * 0.05
    for max_norm in (25.0, 1e9):
        mine = g.clone()
        norm = ops.clip_grad_norm_(mine, max_norm)
        p = torch.nn.Parameter(torch.zeros_like(g))
        p.grad = g.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([p], max_norm)
        np.testing.assert_allclose(norm.item(), ref_norm.item(), rtol=1e-5)
        np.testing.assert_allclose(mine.cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
    bad = g.clone()
    bad[12345] = float("nan")
    kept = bad.clone()
    assert not np.isfinite(ops.clip_grad_norm_(bad, 25.0).item())
    assert torch.equal(torch.nan_to_num(bad), torch.nan_to_num(kept))          # untouched: the caller skips the step
