"""Headline benchmark (BASELINE.json): 3D Mask R-CNN training throughput in patches/s on synthetic 128^3
patches (LIDC-shape config 3), one process per GPU, plus the RoIAlign-3D-backward roofline measured live.

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --gpus N ...      (WORLD_SIZE unset: re-launches itself under torch.distributed.run with N ranks;
                                     exits non-zero when the node has fewer than N GPUs -- never an n_gpus:1 line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (how the driver launches N ranks; --gpus must equal WORLD_SIZE)

A "step" = exec.py:68-74 of the reference: net.train_forward(batch) [incl. H2D of the batch], zero_grad,
backward, (gradient all-reduce over RCCL when N > 1), Adam step; per-GPU batch = 8 patches (weak scaling).
Since round 4 the device half of the step can run as ONE hipGraph replay (training.GraphedTrainStep, `--graph 1`); the default
headline launches the step eagerly (faster by ~4 % at one rank on a fast host, see main()) and times the graphed step as the
`graphed_step` leg of every line (with `--graph 1`: the other way round, `eager_step`).  `exec_equivalent` is the step as exec.py consumes it: host numpy batches
(uploaded behind the previous step by training.DevicePrefetcher), the monitoring read-out (`logger_string`, `boxes`,
`monitor_values`) taken every step, the mask head over the detections run like the reference does (mrcnn.py:1046-1048).
Rank 0 prints ONE JSON line.  `roofline` is the dominant custom kernel (RoIAlign-3D backward on the P2 level):
algorithmic bytes / event-timed duration of the C-ABI op with 48 RoIs of the SURVEY.md 8(d) box distribution on the
level, measured after the timed training loop; `roofline.variants` carries the cache-cold run, the train-realistic
placement, the four-level one-launch form and the op as it ran inside the steps.
`cpu_baseline` times a bounded sample of the same work on the host cores with the CPU oracle (native ops) and
torch-CPU (conv path).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# hipGraph replays of the step need the runtime's graph packet capture OFF (medicaldetectiontoolkit_amd/__init__.py explains); set HERE, before
# torch is imported (importing the package changes nothing in the process)
if "torch" not in sys.modules and os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is None:
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    os.environ["MDT_GRAPH_ENV_BEFORE_HIP"] = "1"
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()   # in-tree MIOpen find-db / kernel cache, must precede the first convolution

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_BPS = 8.0e12   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(cf, anchors, seconds_budget=25.0):
    """Bounded CPU sample of one training step's work for ONE patch, timed on this box's host cores:
      * torch-CPU fp32 forward+backward of the FPN + RPN convolutions (what the reference runs through torch on the CPU),
      * the CPU oracle for the native ops the patch causes: RPN NMS over pre_nms_limit boxes, RoIAlign-3D forward of the
        75 detection RoIs with both pools, forward+backward of the 6 sampled RoIs with (7,7,3) and (14,14,5) on P2,
      * the numpy anchor matching of the reference (gt_anchor_matching over all 449 280 anchors, 3 GT boxes;
        oracle/host_numpy.py) -- done on one host core per batch element in the reference (mrcnn.py:894),
      * torch-CPU classifier + mask heads: forward on the 75 detection RoIs, forward+backward on the 6 sampled RoIs.
    Reported as patches/s (a baseline, not a target: the roofline fraction is the quality figure)."""
    from medicaldetectiontoolkit_amd.models import backbone as bb
    from medicaldetectiontoolkit_amd.models.mrcnn import RPN, Classifier, Mask
    from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
    from oracle import host_numpy, oracle
    from medicaldetectiontoolkit_amd.utils.synthetic_data import nms_boxes, random_boxes_3d
    threads = torch.get_num_threads()
    t_start = time.time()
    conv = NDConvGenerator(cf.dim)
    fpn, rpn = bb.FPN(cf, conv), RPN(cf, conv)
    cls_head, mask_head = Classifier(cf, conv), Mask(cf, conv)
    x = torch.randn([1, 1] + list(cf.patch_size))
    t0 = time.time()
    outs = fpn(x)
    loss = sum(sum(o.float().mean() for o in rpn(p)) for p in outs)
    loss.backward()
    t_conv = time.time() - t0
    rng = np.random.default_rng(0)
    t0 = time.time()
    dets = nms_boxes(rng, cf.pre_nms_limit, dim=3, patch=float(cf.patch_size[0]))
    oracle.gpu_nms(dets, cf.rpn_nms_threshold, True)
    p2 = outs[0].detach().numpy()
    boxes = random_boxes_3d(rng, 75)
    ind = np.zeros(75, np.int32)
    det_cls = oracle.crop_and_resize_forward(p2, boxes, ind, tuple(cf.pool_size))
    det_mask = oracle.crop_and_resize_forward(p2, boxes, ind, tuple(cf.mask_pool_size))
    sampled = {}
    for crop in (tuple(cf.pool_size), tuple(cf.mask_pool_size)):
        c = oracle.crop_and_resize_forward(p2, boxes[:6], ind[:6], crop)
        oracle.crop_and_resize_backward(c, boxes[:6], ind[:6], p2.shape)
        sampled[crop] = c
    t_ops = time.time() - t0
    # anchor matching (numpy, float64) over the full anchor table of this patch size
    # (`anchors`: the float64 table the model built on the device, copied back once outside the timed parts)
    ps = np.asarray(cf.patch_size, dtype=np.float64)
    ctr = rng.uniform(0.3, 0.7, size=(3, 3)) * ps
    half = rng.uniform(4, 12, size=(3, 3))
    gt = np.stack([ctr[:, 0] - half[:, 0], ctr[:, 1] - half[:, 1], ctr[:, 0] + half[:, 0], ctr[:, 1] + half[:, 1],
                   ctr[:, 2] - half[:, 2], ctr[:, 2] + half[:, 2]], 1)
    t0 = time.time()
    host_numpy.anchor_matching(anchors, gt, None, cf.anchor_matching_iou, cf.rpn_train_anchors_per_image, cf.rpn_bbox_std_dev, rng=rng)
    t_match = time.time() - t0
    # heads on the pooled features
    t0 = time.time()
    with torch.no_grad():
        h = cls_head.conv2(cls_head.conv1(torch.from_numpy(det_cls)))
        cls_head.linear_class(h.view(75, -1)), cls_head.linear_bbox(h.view(75, -1))
        m = mask_head.conv4(mask_head.conv3(mask_head.conv2(mask_head.conv1(torch.from_numpy(det_mask)))))
        mask_head.conv5(mask_head.relu(mask_head.deconv(m)))
    xc = torch.from_numpy(sampled[tuple(cf.pool_size)]).requires_grad_(True)
    xm = torch.from_numpy(sampled[tuple(cf.mask_pool_size)]).requires_grad_(True)
    h = cls_head.conv2(cls_head.conv1(xc)).view(6, -1)
    m = mask_head.conv4(mask_head.conv3(mask_head.conv2(mask_head.conv1(xm))))
    m = mask_head.sigmoid(mask_head.conv5(mask_head.relu(mask_head.deconv(m))))
    (cls_head.linear_class(h).mean() + cls_head.linear_bbox(h).mean() + m.mean()).backward()
    t_heads = time.time() - t0
    total = t_conv + t_ops + t_match + t_heads
    return {"value": round(1.0 / total, 4), "unit": "patches/s", "cores": int(threads), "kind": "port",
            "sample": "1 patch %s: torch-CPU FPN+RPN fwd+bwd %.2fs + CPU oracle (NMS n=%d, RoIAlign-3D fwd 75 RoIs x2 pools, fwd+bwd 6 RoIs "
                      "(7,7,3)+(14,14,5) on P2) %.2fs + numpy anchor matching (%d anchors, 3 GT) %.2fs + torch-CPU heads %.2fs; wall %.1fs"
                      % ("x".join(map(str, cf.patch_size)), t_conv, cf.pre_nms_limit, t_ops, anchors.shape[0], t_match, t_heads, time.time() - t_start)}


def cpu_baseline_reference(args, timeout_s=420):
    """`cpu_baseline.kind = "reference"`: the REFERENCE's own models/mrcnn.py `train_forward` + backward + torch.optim.Adam step
    (exec.py:39,68-74) on ONE full batch of the benchmarked configuration, on this box's host cores, in a child process
    (oracle/ref_step_cpu.py: reference files from oracle/_ref/ref_models.tar.gz, the CPU oracle behind the four CUDA-only cuda_functions imports)."""
    import subprocess
    # torch-CPU's 3D convolutions scale badly past a few dozen threads on these hosts (256 threads: 266 s for the step that takes 32 s on the 8
    # cores of the build container): the baseline is run at --cpu-threads (default 16), `cores` says so
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_step_cpu.py"), "--patch", args.patch, "--batch", str(args.batch), "--model", args.model,
           "--threads", str(min(args.cpu_threads, os.cpu_count() or 1))]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("oracle/ref_step_cpu.py rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
    rec = json.loads(lines[-1])
    return {"value": rec["patches_per_s"], "unit": "patches/s", "cores": rec["threads"], "kind": "reference",
            "sample": "ONE full step of the reference's own %s.py on torch-CPU: net.train_forward(batch of %d x %s) + zero_grad + backward + torch.optim.Adam.step "
                      "(exec.py:39,68-74), %.1f s; the four CUDA-only cuda_functions imports served by the CPU oracle (OpenMP); child process wall %.1f s; %s"
                      % (args.model, rec["batch"], "x".join(map(str, rec["patch"])), rec["seconds_per_step"], time.time() - t0, rec["logger_string"])}


def _time_op(fn, launches, warmup=10, pre=None):
    """event-bracketed launches on the current stream (the stream the kernels are launched on); seconds per launch.  `pre` runs before
    every launch OUTSIDE the event bracket (used to put the op's small inputs into the cache state they have inside the step)"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, b in ev:
        if pre is not None:
            pre()
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e-3 for a, b in ev)
    return float(np.mean(t)), float(t[len(t) // 2])


def _pyramid_case(rng, cf, batch, n_per_level):
    """48 sampled RoIs routed to the four levels like the level rule of mrcnn.py:403 would (box side ~ anchor scale of the
    level): utils/synthetic_data.trainlike_rois_3d per level, shuffled row order"""
    from medicaldetectiontoolkit_amd.utils.synthetic_data import trainlike_rois_3d
    per = []
    for li, n in enumerate(n_per_level):
        side = float(cf.rpn_anchor_scales["xy"][li][0])
        tb, ti = trainlike_rois_3d(rng, batch, cf.train_rois_per_image, side, float(cf.patch_size[0]))
        keep = rng.permutation(len(tb))[:n]
        per.append((tb[keep], ti[keep], np.full(n, li, dtype=np.int32)))
    order = rng.permutation(sum(n_per_level))
    return (np.concatenate([q[0] for q in per])[order], np.concatenate([q[1] for q in per])[order],
            np.concatenate([q[2] for q in per])[order])


def roialign_bwd_roofline(cf, batch, dev, in_step_prof, in_step_prof48=None, launches=60):
    """Roofline of the dominant custom kernel, RoIAlign-3D backward: algorithmic bytes (every gradient
    map written once + the pooled gradients read once + 28 B per RoI) / event-timed duration of the C-ABI op, measured
    after the training loop.  pool (14,14,5), N = 48 valid RoIs.
      headline   P2 map 8x36x32x32x128, the survey's own box distribution (centre U(0,1)^3, xy side
                 log-uniform 8..64 px, z 2..16 px, box_ind ~ U{0..B-1}), all 48 RoIs on the level, the SAME output
                 buffer rewritten every launch (cache-warm: the 256 MiB Infinity Cache may absorb part of the write);
      variants   the same case CACHE-COLD (4 output buffers = 604 MB rotated), the train-realistic placement (6 sampled
                 RoIs per element around one object, level-sized boxes) warm and cold, all four pyramid levels in ONE launch
                 with the 48 RoIs routed 24/12/8/4 (warm and cold), and the op as it ran inside the timed training steps."""
    from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
    from medicaldetectiontoolkit_amd.utils.synthetic_data import random_boxes_3d, trainlike_rois_3d
    shapes = [(batch, cf.end_filts) + tuple(int(v) for v in sh) for sh in cf.backbone_shapes]
    shape = shapes[0]
    crop = tuple(cf.mask_pool_size)
    n = cf.train_rois_per_image * batch
    V, P = int(np.prod(shape[2:])), int(np.prod(crop))
    alg = 4.0 * batch * cf.end_filts * V + 4.0 * n * cf.end_filts * P + 28.0 * n
    alg_pyr = 4.0 * sum(int(np.prod(sh)) for sh in shapes) + 4.0 * n * cf.end_filts * P + 36.0 * n
    rng = np.random.default_rng(0)
    g = torch.randn((n, cf.end_filts) + crop, device=dev)
    tb, ti = trainlike_rois_3d(rng, batch, cf.train_rois_per_image, 8.0, float(cf.patch_size[0]))
    rb, ri = random_boxes_3d(rng, n), rng.integers(0, batch, size=n).astype(np.int32)
    pb, pi, pl = _pyramid_case(rng, cf, batch, (n // 2, n // 4, n // 6, n - n // 2 - n // 4 - n // 6))
    traffic = {}
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc", "traffic.json")))
    except Exception:
        pass
    n_rot = 4                      # 4 x 151 MB = 604 MB > the 256 MiB Infinity Cache: every launch writes lines it does not hold
    rot = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(n_rot)]
    rot_pyr = [[rot[k]] + [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes[1:]] for k in range(n_rot)]
    state = {"k": 0}

    def rec_of(mean_s, med_s, byts, rois, tkey=None):
        r = {"achieved": round(byts / mean_s / 1e9, 1), "frac": round(byts / mean_s / HBM_PEAK_BPS, 4), "avg_us": round(mean_s * 1e6, 2),
             "median_us": round(med_s * 1e6, 2), "launches": launches, "alg_bytes_per_launch": int(byts), "rois": rois}
        t = traffic.get(tkey) if tkey else None
        r["traffic"] = t["hbm_bytes"] if t else None
        return r

    def single(boxes, ind, cold, tkey=None, touch=False):
        bx, bi = torch.from_numpy(boxes).to(dev), torch.from_numpy(ind).to(dev)

        def fn():
            state["k"] += 1
            _roi_align_impl.crop_backward(g, bx, bi, shape, out=rot[state["k"] % n_rot if cold else 0])

        def pre():       # what the step does right before this op: the mask head's backward has just WRITTEN g, the boxes were just read
            g.add_(0.0)
            bx.add_(0.0)
            bi.add_(0)
        return rec_of(*_time_op(fn, launches, pre=pre if touch else None), alg, n, tkey)

    def pyramid(cold):
        bx, bi, lv = torch.from_numpy(pb).to(dev), torch.from_numpy(pi).to(dev), torch.from_numpy(pl).to(dev)

        def fn():
            state["k"] += 1
            _roi_align_impl.pyramid_backward(g, bx, bi, lv, shapes, outs=rot_pyr[state["k"] % n_rot if cold else 0])
        return rec_of(*_time_op(fn, launches), alg_pyr, n)

    def plain_fill(cold):
        """calibration: what a bare 151 MB fill (torch's fill kernel, no RoIs, no skip logic) does under the same protocol"""
        def fn():
            state["k"] += 1
            rot[state["k"] % n_rot if cold else 0].zero_()
        mean_s, med_s = _time_op(fn, launches)
        byts = 4.0 * batch * cf.end_filts * V
        return {"achieved": round(byts / mean_s / 1e9, 1), "frac": round(byts / mean_s / HBM_PEAK_BPS, 4), "avg_us": round(mean_s * 1e6, 2),
                "median_us": round(med_s * 1e6, 2), "launches": launches, "bytes": int(byts)}

    warm = single(rb, ri, False, "survey_random_48_rois")
    # HEADLINE (round 4, ADVICE r3 / VERDICT r3 item 3): the cache state the op has inside a training step -- the 151 MB gradient map
    # is fresh memory (4 rotating output buffers = 604 MB > the 256 MiB Infinity Cache: every launch writes lines the cache does not
    # hold), while the pooled gradients `g` and the boxes were produced microseconds earlier (touched right before the launch,
    # outside the event bracket).  The same-buffer (cache-warm) and everything-cold figures are variants.
    head = single(rb, ri, True, "survey_random_48_rois", touch=True)
    variants = {
        "P2_survey_8d_random_boxes_same_output_buffer_cache_warm": warm,
        "P2_survey_8d_random_boxes_cache_cold": single(rb, ri, True),
        "P2_train_realistic_placement": single(tb, ti, False, "trainlike_48_rois"),
        "P2_train_realistic_placement_step_cache_state": single(tb, ti, True, touch=True),
        "P2_train_realistic_placement_cache_cold": single(tb, ti, True),
        "all_four_levels_one_launch_48_rois": pyramid(False),
        "all_four_levels_one_launch_48_rois_cache_cold": pyramid(True),
        "calibration_torch_zero_fill_151MB": plain_fill(False),
        "calibration_torch_zero_fill_151MB_cache_cold": plain_fill(True),
    }
    # the op as it ran inside the timed steps: ONE launch for all four pyramid levels (mdt_pyramid_roi_align_backward), so
    # the algorithmic bytes are the four gradient maps + the pooled gradients of the RoIs the level rule kept; fresh output
    # maps from the allocator after 50 ms of convolutions: cache-cold by construction
    for key, pr in (("in_training_step_all_levels_one_launch", in_step_prof),
                    ("in_training_step_all_levels_one_launch_rois_heads_full", in_step_prof48)):
        recs = [(a.elapsed_time(b) * 1e-3, m) for a, b, m in (pr or []) if m.get("mode") == "pyramid" and m["crop"] == crop]
        if recs:
            maps_bytes = 4.0 * sum(int(np.prod(sh)) for sh in recs[0][1]["levels"])
            byts = [maps_bytes + 4.0 * int(m["n_valid"].item()) * cf.end_filts * P + 36.0 * m["n_rows"] for _, m in recs]
            dur = float(np.mean([d for d, _ in recs]))
            variants[key] = {
                "achieved": round(float(np.mean(byts)) / dur / 1e9, 1), "frac": round(float(np.mean(byts)) / dur / HBM_PEAK_BPS, 4),
                "avg_us": round(dur * 1e6, 2), "launches": len(recs), "alg_bytes_per_launch": int(np.mean(byts)),
                "rois": round(float(np.mean([int(m["n_valid"].item()) for _, m in recs])), 2)}
    kernel_desc = ("crop_bwd_gather_kernel (mdt_crop_and_resize_3d_backward, csrc/roi_align_bwd_v3.hip): P2 %s, pool %s, %d RoIs on the level, "
                   "SURVEY 8(d) box distribution, training-step cache state (output map = fresh memory: 4 x 151 MB rotated; pooled gradients and "
                   "boxes just produced)" % ("x".join(map(str, shape)), "x".join(map(str, crop)), n))
    variants["P2_survey_8d_random_boxes_step_cache_state"] = dict(head)
    full = variants.get("in_training_step_all_levels_one_launch_rois_heads_full")
    if full is not None and full.get("rois", 0) >= 1.0:        # (the RoI count drifts while the weights train on the one batch; it is stated in the record -- the
                                                                # four gradient maps are 98 % of the launch's bytes either way)
        # HEADLINE: the op AS IT RUNS in the training step -- the mask head's pyramid backward (mdt_pyramid_roi_align_backward: all four
        # gradient maps of the batch in ONE launch, 173.7 MB + the pooled gradients), event-timed inside eager training steps on a batch
        # whose GT boxes come from the net's own proposals, so the RoI heads are (nearly) full instead of ~8 valid RoIs of 48: the real
        # allocator / cache state, the real box distribution of the level rule.  The synthetic single-level cases are variants.
        head = dict(full)
        head["traffic"] = None
        head["median_us"] = None
        try:        # PMC bytes of THIS launch in training steps with full RoI heads (tools/r05_instep_profile.sh, offline)
            tdir = "r06" if os.path.exists(os.path.join(ROOT, "profiles", "r06", "traffic_instep.json")) else "r05"
            ti = json.load(open(os.path.join(ROOT, "profiles", tdir, "traffic_instep.json")))["in_training_step_heads_full"]
            head["traffic"] = ti["hbm_bytes"]
            head["traffic_detail"] = {k: ti.get(k) for k in ("write_kb", "fetch_kb", "fill_write_kb_calibration", "algorithmic_bytes", "valid_rois",
                                                                "rocprofv3_kernel_us_mean")}
            instep_traffic_source = ("OFFLINE measurement, not part of this run: profiles/" + tdir + "/traffic_instep.json -- rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in "
                                     "separate passes over tools/instep_heads_full.py (training steps with full RoI heads, the mask head's pyramid backward launch), "
                                     "FETCH_SIZE doubled (gfx950), WRITE_SIZE calibrated on a 150 994 944-byte fill of the same pass (tools/instep_profile.sh)")
        except Exception:
            instep_traffic_source = None
        kernel_desc = ("crop_bwd_gather_kernel (mdt_pyramid_roi_align_backward, csrc/roi_align_bwd_v3.hip) AS IT RAN INSIDE TRAINING STEPS: the mask head's "
                       "backward, all four pyramid gradient maps (8 x 36 x {32x32x128, 16x16x64, 8x8x32, 4x4x16}) in one launch, pool %s, %.1f valid RoIs of %d "
                       "(GT boxes derived from the net's own proposals so that the RoI heads are full), HIP events around the launch in eager steps"
                       % ("x".join(map(str, crop)), full["rois"], n))
    out = {"bound": "hbm", "achieved": head["achieved"], "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s", "frac": head["frac"], "traffic": head.get("traffic"),
           "traffic_source": (locals().get("instep_traffic_source") or
                              ("OFFLINE measurement, not part of this run: profiles/r03_pmc/traffic.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes "
                               "of this op, tools/gpu_pmc.sh)")) if head.get("traffic") else None,
           "traffic_detail": head.get("traffic_detail"),
           "kernel": kernel_desc,
           "alg_bytes_per_launch": head["alg_bytes_per_launch"], "avg_us": head["avg_us"], "median_us": head.get("median_us"), "launches": head.get("launches", launches),
           "timing": "HIP events around every launch on the launch stream; adds ~2 us over the kernel's own duration (profiles/r03_* rocprofv3 stats)",
           "variants": variants}
    return out


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this file as N ranks (one per GPU) under torch.distributed.run
    on 127.0.0.1 and hand its exit code back."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd))


def _graph_preflight(args, device_index, legs=False, timeout_s=600):
    """hipGraph work in a CHILD process (see graph_preflight_main): `legs=False` -- capture + three replays, as the check that precedes a
    graphed headline; `legs=True` -- the graphed A/B legs of an eager headline.  Returns (ok, note, legs dict or None)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-preflight", "legs" if legs else "check", "--patch", args.patch, "--batch", str(args.batch),
           "--gmax", str(args.gmax), "--steps", str(args.steps), "--device-index", str(device_index), "--channels-last", str(args.channels_last),
           "--merge-rpn-heads", str(args.merge_rpn_heads), "--sparse-rpn-loss", str(args.sparse_rpn_loss), "--step-form", args.step_form] + (
               ["--no-exec-leg"] if args.no_exec_leg else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                          "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    t0 = time.time()
    out = None
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        ok = "GRAPH_PREFLIGHT_OK" in r.stdout and (r.returncode == 0 or not legs)
        for line in r.stdout.splitlines():
            if line.startswith("GRAPH_LEGS "):
                out = json.loads(line[len("GRAPH_LEGS "):])
        ok = ok and (out is not None or not legs)
        note = "ok (%.0f s)" % (time.time() - t0) if ok else "FAILED rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])
    except subprocess.TimeoutExpired:
        ok, note = False, "FAILED: timeout after %d s" % timeout_s
    return ok, note, out


def graph_preflight_main(args):
    """child process (`--graph-preflight [legs]`): ALL hipGraph work of a `--graph 0` bench line happens here, the capture first thing in a
    fresh process -- a capture AFTER eager training steps of the same net in one process segfaults in hipStreamEndCapture on this stack
    (the AccumulateGrad nodes of the parameters were created on the default stream by the eager backward and pull it into the capture;
    r04_final first try), and no runtime crash may ever cost the headline line.  Prints GRAPH_PREFLIGHT_OK and, with `legs`, one JSON line:
    the graphed step's rate + host-work breakdown, hipGraphLaunch host time with the GPU idle, the exec-equivalent leg in graphed form."""
    from medicaldetectiontoolkit_amd import training
    from medicaldetectiontoolkit_amd.configs import Configs
    from medicaldetectiontoolkit_amd.models import mrcnn
    from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device
    idx = args.device_index or 0
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    torch.backends.cudnn.benchmark = True
    patch = [int(v) for v in args.patch.split(",")]
    cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=args.batch, channels_last=bool(args.channels_last))
    mrcnn.MERGE_RPN_HEADS = bool(args.merge_rpn_heads)
    mrcnn.SPARSE_RPN_LOSS = bool(args.sparse_rpn_loss)
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=dev)
    opt = training.build_optimizer(net, cf, flat=True)
    exec_form = args.step_form == "exec"
    if exec_form:
        cf.run_detection_mask_head_in_training = True
    step = training.GraphedTrainStep(net, opt, gmax=args.gmax, monitor="deferred" if exec_form else False, with_masks=exec_form)
    pool = [to_device(make_batch(patch, args.batch, seed=1000 + i), dev) for i in range(3)]
    for i in range(3):
        res = step(pool[i % 3])
    torch.cuda.synchronize()
    if not np.isfinite(float(res["torch_loss"])):
        raise SystemExit("graph preflight: non-finite loss")
    print("GRAPH_PREFLIGHT_OK loss=%.4f" % float(res["torch_loss"]), flush=True)
    if args.graph_preflight != "legs":
        return
    n = max(2, min(args.steps, 10))
    step.host_ms = {}
    t0 = time.time()
    used = 0
    for i in range(n):
        r = step(pool[i % 3])
        if "logger_string" in r:
            used += len(r["logger_string"]) + len(r["boxes"]) + len(r["monitor_values"])
    torch.cuda.synchronize()
    dt = time.time() - t0
    hm, step.host_ms = step.host_ms, None
    c = max(1, hm.pop("calls", 1))
    hw = {k: round(v / c, 3) for k, v in hm.items()}
    hw["work_total"] = round(sum(v for k, v in hw.items() if k != "ring_wait_backpressure"), 3)
    idle = []
    for i in range(3):               # hipGraphLaunch with the GPU idle: the host cost of a replay itself
        torch.cuda.synchronize()
        t1 = time.time()
        step.graph.replay()
        idle.append((time.time() - t1) * 1e3)
    torch.cuda.synchronize()
    rec = {"graphed_step": {"value": round(args.batch * n / dt, 3), "unit": "patches/s", "steps": n, "ms_per_step": round(dt / n * 1e3, 2),
                            "host_ms_per_step": hw, "hipGraphLaunch_host_ms_gpu_idle": round(min(idle), 2),
                            "step_form": args.step_form,
                            "note": "training.GraphedTrainStep in a child process (fresh net of the same seed, same batch generator; the headline's step form): the device half of the step as "
                                    "ONE hipGraph replay (+ Adam launch); `replay` in host_ms_per_step includes waiting for the previous replay of the same "
                                    "graph (= the GPU), the idle figure is the launch itself"}}
    if not args.no_exec_leg:
        try:
            rec["exec_equivalent_graphed"] = exec_equivalent_leg(net, opt, cf, patch, args, dev, True)
        except Exception as e:
            rec["exec_equivalent_graphed"] = {"failed": repr(e)[:300]}
    print("GRAPH_LEGS " + json.dumps(rec), flush=True)


def exec_equivalent_leg(net, opt, cf, patch, args, dev, use_graph, deferred=True):
    """exec.py:67-79 as the reference runs it: `batch = next(batch_gen)` (host numpy), `results = net.train_forward(batch)`, zero_grad,
    backward, step, then the consumers of results_dict (`logger_string` logged, `boxes` appended for the training metrics,
    `monitor_values` plotted) -- with the mask head over the detections on (mrcnn.py:1046-1048).  The batch stream goes through
    training.DevicePrefetcher; the read-out is one packed device->host copy per step.  deferred=True (round 5): that copy is asynchronous and
    the entries of step i are consumed while step i + 1 is already queued (monitor="deferred": exec.py's log line / box list arrive one batch
    late, no host sync in the step); deferred=False: the synchronous read-out of round 4 (1 sync per step)."""
    from medicaldetectiontoolkit_amd import training
    from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch
    n = max(3, args.steps if deferred else min(args.steps, 8))
    host_pool = [make_batch(patch, args.batch, seed=50 + i) for i in range(2)]
    seq = [host_pool[i % 2] for i in range(n + 2)]
    prev = getattr(cf, "run_detection_mask_head_in_training", False)
    cf.run_detection_mask_head_in_training = True
    try:
        mode = "deferred" if deferred else True
        if use_graph:
            xstep = training.GraphedTrainStep(net, opt, gmax=args.gmax, monitor=mode, with_masks=True)
        else:
            def xstep(b):
                return training.train_step(net, opt, b, monitor=mode)
        pf = training.DevicePrefetcher(seq, dev)
        consumed = 0
        last_log = None
        for _ in range(2):                       # capture / warm-up
            res = xstep(next(pf))
        torch.cuda.synchronize()
        if use_graph:
            xstep.host_ms = {}
        t_wait = 0.0
        t0 = time.time()
        while True:
            tw = time.time()
            b = next(pf, None)
            t_wait += time.time() - tw
            if b is None:
                break
            res = xstep(b)
            if "logger_string" in res:       # (deferred: absent on the very first call only)
                consumed += len(res["logger_string"]) + len(res["boxes"]) + len(res["monitor_values"])     # what exec.py:76-79 reads every batch
                last_log = res["logger_string"]
        torch.cuda.synchronize()
        dt = time.time() - t0
    finally:
        cf.run_detection_mask_head_in_training = prev
    host_ms = None
    if use_graph and xstep.host_ms:
        c = max(1, xstep.host_ms.pop("calls", 1))
        host_ms = {k: round(v / c, 2) for k, v in xstep.host_ms.items()}
        host_ms["wait_for_next_batch"] = round(t_wait / c * 1e3, 2)
    return {"value": round(args.batch * n / dt, 3), "unit": "patches/s", "steps": n, "ms_per_step": round(dt / n * 1e3, 2),
            "device_to_host_syncs_per_step": 0 if deferred else 1, "readout": "deferred by one step (asynchronous copy)" if deferred else "synchronous",
            "graph": bool(use_graph), "host_ms_per_step": host_ms, "last_logger_string": last_log,
            "note": "host numpy batches (training.DevicePrefetcher: upload of batch i+1 behind step i) + monitoring read-out every step (one packed D2H"
                    + (", asynchronous: the entries of step i are consumed after step i+1 was queued" if deferred else "") +
                    ") + mask head over the detections (mrcnn.py:1046-1048) + box lists built on the host"}


def secondary_configs(timeout_s=240):
    """BASELINE configs 1, 2 and 5 under the same clock as the headline line: each runs in its OWN process with a
    timeout (a cold MIOpen find for their layer shapes must never hold the headline line back), results parsed from the
    child's JSON line.  config 2: LIDC-shape 3D Retina U-Net, 128^3, batch 8, 3 timed steps; config 5: one 512x512x256
    volume, 75 patches, bf16, single pass, device-resident predictor (tools/bench_inference.py)."""
    import subprocess
    env = dict(os.environ)
    env["MDT_MIOPEN_SKIP_NAIVE"] = "1"          # never let the find try the naive direct solvers on 128^3 decoder maps
    out = {}
    jobs = {
        "config2_retina_unet_128_b8": [sys.executable, os.path.abspath(__file__), "--model", "retina_unet", "--steps", "3", "--warmup", "2", "--settle", "2",
                                       "--no-cpu-baseline", "--no-h2d-leg", "--no-rccl-selftest", "--no-secondary", "--no-roofline"],
        "config1_toy2d_retina_net_64x64_b20": [sys.executable, os.path.join(ROOT, "tools", "bench_toy2d.py"), "--steps", "30", "--sizes", "64"],
        "config5_inference_512x512x256_bf16": [sys.executable, os.path.join(ROOT, "tools", "bench_inference.py"), "--amp", "bf16",
                                               "--test-aug", "0", "--repeats", "1"],
    }
    for key, cmd in jobs.items():
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            rec = json.loads(lines[-1]) if lines else {"failed": "no JSON line; rc=%d; %s" % (r.returncode, r.stderr[-300:])}
        except subprocess.TimeoutExpired:
            rec = {"failed": "timeout after %d s (cold MIOpen find?)" % timeout_s}
        except Exception as e:
            rec = {"failed": repr(e)}
        keep = ("metric", "value", "unit", "ms_per_step", "steps", "config", "images_per_s", "loss_first5_mean", "loss_last5_mean", "finite", "patients_per_min", "patches_per_s", "s_per_patient", "n_patches",
                "n_passes", "forwards", "raw_boxes", "boxes_after_wbc", "amp", "failed")
        out[key] = {k: rec[k] for k in keep if k in rec}
        out[key]["wall_s"] = round(time.time() - t0, 1)
    return out


def rccl_world1_selftest(net, opt, batch, dev, steps=3, use_graph=False, gmax=8):
    """The N > 1 gradient path on its real backend, on a 1-GPU box: a world-size-1 `nccl` (= RCCL) process group, the
    collectives of training.FlatGradAllReduce forced on.  (1) local gradient (hooks off) vs the same buffer after the
    bucket all-reduces: must be bit-identical (sum over one rank, / 1); (2) `steps` train_steps with the async bucket
    all-reduces launched from the backward hooks: parameters stay finite, step time reported."""
    from medicaldetectiontoolkit_amd import training
    t_init = time.time()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=dev)
    try:
        sync = training.FlatGradAllReduce(net, force=True)
        if isinstance(opt, training.FlatAdam):      # its gradient views must be the collective's buffer: a fresh one over `sync`
            g = opt.param_groups[0]
            opt = training.FlatAdam(net.parameters(), lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], grad_sync=sync)
        res = net.train_forward(batch, monitor=False)
        sync.zero()
        sync.force = False
        res["torch_loss"].backward()
        local = sync.flat.clone()
        sync.force = True
        sync.finish()
        torch.cuda.synchronize()
        identical = bool(torch.equal(sync.flat, local))
        opt.step()
        t_init = time.time() - t_init
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            training.train_step(net, opt, batch, grad_sync=sync, monitor=False)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / steps * 1e3
        launched = sync._next
        rec_graph = None
        if use_graph:      # the N > 1 form of the graphed step: gradients accumulate into the flat buffer inside the graph, the bucket
            try:           # all-reduces and the Adam launch follow the replay
                gs = training.GraphedTrainStep(net, opt, grad_sync=sync, gmax=gmax)
                for _ in range(2):
                    gs(batch)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(steps):
                    gs(batch)
                torch.cuda.synchronize()
                rec_graph = {"ms_per_step_with_collectives": round((time.time() - t0) / steps * 1e3, 2), "buckets_all_reduced_per_step": int(sync._next)}
            except Exception as e:
                rec_graph = {"failed": repr(e)[:200]}
        flat_params = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        return {"backend": dist.get_backend(), "world": dist.get_world_size(), "buckets_all_reduced_per_step": int(launched),
                "grad_bit_identical_after_allreduce": identical, "params_finite": bool(torch.isfinite(flat_params).all()),
                "ms_per_step_with_collectives": round(ms, 2), "graphed_step": rec_graph, "init_plus_first_step_s": round(t_init, 2)}
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patch", type=str, default="128,128,128")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--model", type=str, default="mrcnn", choices=["mrcnn", "retina_unet"],
                    help="mrcnn = BASELINE config 3 (headline); retina_unet = config 2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the cpu_baseline's reference step (oracle/ref_step_cpu.py)")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 2 and 5 (run in child processes after the headline line is assembled, N = 1 only)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the RoIAlign-backward roofline section (child runs of --secondary)")
    ap.add_argument("--no-rccl-selftest", action="store_true", help="skip the world-size-1 RCCL self-test after the timed loop")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the extra host-batch steps after the timed loop (profiling runs)")
    ap.add_argument("--graph", type=int, default=0, help="0 (default): the headline steps are launched eagerly and the graphed step is timed as the `graphed_step` leg; 1: the headline steps are ONE hipGraph replay each (training.GraphedTrainStep) and the eager step is the `eager_step` leg")
    ap.add_argument("--no-graph-leg", action="store_true", help="with --graph 0: skip the graphed A/B leg and the graphed exec_equivalent form")
    ap.add_argument("--gmax", type=int, default=8, help="GT objects per batch element the fixed-size GT table of the graphed step holds")
    ap.add_argument("--no-graph-preflight", action="store_true", help="skip the child-process capture check that precedes the graphed run")
    ap.add_argument("--graph-preflight", type=str, default="", help=argparse.SUPPRESS)
    ap.add_argument("--device-index", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-eager-leg", action="store_true", help="skip the eager A/B steps after the timed loop")
    ap.add_argument("--no-dense-rpn-leg", action="store_true", help="skip the A/B steps with the RPN losses differentiated through the dense RPN outputs")
    ap.add_argument("--no-exec-leg", action="store_true", help="skip the exec.py-equivalent leg (host batches + monitoring read-out + mask head over detections)")
    ap.add_argument("--fused-adam", type=int, default=0)
    ap.add_argument("--flat-adam", type=int, default=1, help="1 (default): training.FlatAdam -- parameters, gradients and Adam moments in flat buffers, the update one launch of csrc/adam.hip; 0: torch.optim.Adam (A/B)")
    ap.add_argument("--fused-epilogue", type=int, default=1, help="1 (default): fused bias/residual/ReLU conv epilogues (csrc/epilogue.hip); 0: torch ops (A/B)")
    ap.add_argument("--conv-bwd-as-fwd", type=int, default=1, help="1 (default): input gradients of unit-stride convolutions as forward convolutions (utils/fused_epilogue._ConvStride1); 0: MIOpen backward-data (A/B)")
    ap.add_argument("--stem-s2d", type=int, default=1, help="1 (default): stem convolution forward in space-to-depth form (utils/fused_epilogue._ConvStem221); 0: as is (A/B)")
    ap.add_argument("--wgrad-1x1", type=int, default=1, help="1 (default): weight gradients of the 1x1x1 convolutions with the fp32-MFMA kernel (csrc/conv1x1_wgrad.hip); 0: MIOpen backward-weights (A/B)")
    ap.add_argument("--stem-wgrad", type=int, default=1, help="1 (default): weight gradient of the one-channel 7x7x7 stem on the fp32-MFMA kernel (csrc/conv_stem_wgrad.hip); 0: MIOpen (A/B)")
    ap.add_argument("--stem-fwd", type=int, default=1, help="1 (default): forward of the one-channel 7x7x7 stem on the fp32-MFMA kernel (csrc/conv_stem_fwd.hip); 0: MIOpen, space-to-depth form (A/B)")
    ap.add_argument("--conv-s2d", type=int, default=1, help="1 (default): stride-(2, 2, 1) layers with many input channels (the Retina U-Net's C1) in space-to-depth form, weight gradient on the fp32-MFMA kernel of csrc/conv_s221.hip; 0: MIOpen's direct problem (A/B)")
    ap.add_argument("--conv-s221-fwd", type=int, default=1, help="1 (default): forward of the stride-(2, 2, 1) many-channel layer (the Retina U-Net's C1) on the fp32-MFMA kernel of csrc/conv_s221.hip; 0: MIOpen on the space-to-depth problem (A/B)")
    ap.add_argument("--conv-s221-dgrad", type=int, default=1, help="1 (default): input gradient of the same layer on this repo's fp32-MFMA kernel; 0: MIOpen forward convolution on the padded output gradient + fold (A/B)")
    ap.add_argument("--conv-win", type=int, default=1, help="1 (default): the few-channel 3x3x3 layers' forward / input gradient on the unit-stride window kernel where supported; 0: csrc/conv3x3x3_small.hip (A/B)")
    ap.add_argument("--conv-win-wgrad", type=int, default=1, help="1 (default): weight gradient of the few-channel 3x3x3 layers on the unit-stride window kernel (mdt_conv_win_wgrad); 0: csrc/conv3x3x3_small.hip / MIOpen (A/B)")
    ap.add_argument("--conv3-small", type=int, default=1, help="1 (default): the few-channel 3x3x3 convolutions (18 -> 18 on the large maps) on the fp32-MFMA kernel (csrc/conv3x3x3_small.hip), forward and input gradient; 0: MIOpen (A/B)")
    ap.add_argument("--conv3-small-epilogue", type=int, default=int(os.environ.get("MDT_C3_EPILOGUE", "1")), help="1 (default): bias + ReLU of the few-channel 3x3x3 layers inside the convolution kernel's epilogue (utils/fused_epilogue._Conv3SmallBiasReLU); 0: separate epilogue pass (A/B)")
    ap.add_argument("--res-tap", type=int, default=1, help="1 (default): identity ResBlocks produce their input gradient already added to the residual gradient (utils/fused_epilogue._Conv1x1ResTap, csrc/epilogue.hip); 0: conv backward + autograd's accumulation pass (A/B)")
    ap.add_argument("--head-as-linear", type=int, default=1, help="1 (default): the classifier head's full-extent / 1x1x1 convolutions as GEMMs (models/mrcnn.py Classifier); 0: MIOpen convolutions (A/B)")
    ap.add_argument("--merge-rpn-heads", type=int, default=1, help="1 (default): conv_class and conv_bbox of the RPN as one 1x1 convolution over the shared 128-channel map (models/mrcnn.py RPN); 0: two layers (A/B)")
    ap.add_argument("--upsample-nearest-cl", type=int, default=1, help="1 (default): the FPN's nearest x2 up-sampling on channels-last storage (utils/fused_epilogue._UpsampleNearestCL); 0: F.interpolate with its layout conversions (A/B)")
    ap.add_argument("--stride-tap", type=int, default=1, help="1 (default): a stage output is sub-sampled once for the two strided 1x1 layers of the next stage and the three gradients meet in one node (utils/fused_epilogue._StrideTap); 0: three autograd consumers (A/B)")
    ap.add_argument("--sparse-rpn-loss", type=int, default=1, help="1 (default): the RPN losses differentiate through the 48 sampled anchors only (models/mrcnn.rpn_at_anchors; the dense RPN forward carries no graph); 0: through the dense outputs like the reference (A/B)")
    ap.add_argument("--upsample-cl", type=int, default=1, help="1 (default): channels-last x2 (y, x) linear up-sampling kernel of the Retina U-Net decoder (csrc/upsample.hip); 0: torch (A/B)")
    ap.add_argument("--roialign-cl", type=int, default=1, help="1 (default): the RoI heads pool the channels-last pyramid maps as they are (mdt_pyramid_roi_align_forward_cl); 0: one row-major copy of the pyramid per forward (A/B)")
    ap.add_argument("--fused-glue", type=int, default=1, help="1 (default): level rule, RPN sampling, box targets, detection target layer, refine_detections and the sampled-anchor gather as single launches of csrc/glue.hip; 0: chains of small tensor operations (A/B)")
    ap.add_argument("--bias-grad-in-launch", type=int, default=0, help="1: the bias gradient's second stage inside the backward epilogue's launch (measured slower, profiles/r06/r06_bias_grad_in_launch_probe.txt); 0 (default): separate finish launch")
    ap.add_argument("--conv1x1-fwd", type=int, default=1, help="1 (default): the bottleneck 1x1 layers of the C2 / C3 stages with bias / residual / ReLU in one pass (mdt_conv1x1_forward); 0: MIOpen + epilogue kernel (A/B)")
    ap.add_argument("--conv1x1-bwd", type=int, default=1, help="1 (default): ReLU mask + bias gradient + input gradient of conv3 in the C2 blocks in one pass (mdt_conv1x1_backward); 0: epilogue kernel + CK (A/B)")
    ap.add_argument("--rpn-heads-fused", type=int, default=1, help="1 (default): the dense RPN forward runs both heads on the raw conv_shared output in one launch per level (mdt_rpn_heads_forward); 0: per-level modules + torch.cat (A/B)")
    ap.add_argument("--stem-pool-fused", type=int, default=1, help="1 (default): stem + bias + ReLU + max pooling as one autograd node, ReLU mask / bias gradient at the pooled resolution; 0: two nodes (A/B)")
    ap.add_argument("--conv-c0", type=int, default=1, help="1 (default): the one-channel 3x3x3 first layer of the stride-1 backbone (Retina U-Net C0[0]) on this repo's kernels (csrc/conv_c0.hip); 0: MIOpen + layout conversions (A/B)")
    ap.add_argument("--seg-head-composed", type=int, default=1, help="1 (default): the Retina U-Net's final_conv o P0_conv2 as one 36 -> 2 3x3x3 layer on csrc/conv_seg.hip; 0: the two layers as they are (A/B)")
    ap.add_argument("--bias-bwd-no-copy", type=int, default=1, help="1 (default): bias-only layers' backward returns the output gradient itself; 0: stores a copy (A/B)")
    ap.add_argument("--bias-grad-transpose", type=int, default=1, help="1 (default): row-major output gradients of channels-last bias-only layers converted + reduced in one pass (A/B)")
    ap.add_argument("--lateral-upsample-fused", type=int, default=1, help="1 (default): the FPN's top-down add reads the coarser map directly; 0: materialised up-sampling (A/B)")
    ap.add_argument("--flip-batched", type=int, default=1, help="1 (default): all flipped filters of a step from one launch; 0: one launch per layer (A/B)")
    ap.add_argument("--shared-pyramid-grad", type=int, default=1, help="1 (default): one gradient buffer per pyramid map for the two RoIAlign backward launches and the sampled-anchor RPN scatter (PyramidGradAccumulator); 0: three dense gradients added by autograd (A/B)")
    ap.add_argument("--pool-cl", type=int, default=1, help="1 (default): channels-last max pooling kernel of the stem (csrc/pool.hip); 0: torch (A/B)")
    ap.add_argument("--pin-cores", type=int, default=1, help="N > 1: 1 (default) pins every rank to its own slice of the cores of its GPU's NUMA node (utils/affinity.py); 0: only caps the intra-op threads")
    ap.add_argument("--backend", type=str, default="nccl", help="nccl (= RCCL, default) | gloo (debug: lets several ranks share one GPU)")
    ap.add_argument("--channels-last", type=int, default=1)
    ap.add_argument("--settle", type=int, default=-1, help="untimed steps after the --warmup steps (default: as many as bring warm-up to 25 steps; 0 to switch off)")
    ap.add_argument("--step-form", type=str, default="exec", choices=["exec", "no-readout"],
                    help="exec (default): the step exec.py:68-79 runs -- train_forward WITH the per-batch read-out (logger_string, box lists, monitor_values; one "
                         "packed asynchronous device->host copy, consumed every step) and the mask head over the detections (mrcnn.py:1046-1048), backward, Adam; "
                         "no-readout: round 5's headline form (train_forward(monitor=False), no detection mask head), otherwise timed as the `no_readout_step` leg")
    ap.add_argument("--host-batches", action="store_true", help="hand numpy batches to train_forward (PCIe-inclusive rate)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    if args.graph_preflight:
        return graph_preflight_main(args)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    n_dev = torch.cuda.device_count()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if n_dev < args.gpus and args.backend == "nccl":
            raise SystemExit("bench.py: --gpus %d requested but this node exposes %d GPU(s); refusing to report a smaller run" % (args.gpus, n_dev))
        _self_launch(args.gpus)
    # stdout carries the ONE JSON line and nothing else: libraries that write to fd 1 from C (the RCCL version banner, MIOpen
    # notes; some only at exit, i.e. AFTER the line) are sent to stderr for the lifetime of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    if args.backend == "nccl" and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > n_dev:
        raise SystemExit("bench.py: %s local ranks but only %d GPU(s): RCCL needs one GPU per rank (use --backend gloo to share a GPU for debugging)"
                         % (os.environ.get("LOCAL_WORLD_SIZE", world), n_dev))
    local_dev = local_rank % n_dev      # == local_rank whenever there is one GPU per rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    affinity_rec = None
    if world > 1:   # N processes share the host: each rank gets a disjoint slice of the cores of its GPU's NUMA node (utils/affinity.py)
        from medicaldetectiontoolkit_amd.utils import affinity
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if args.pin_cores:
            affinity_rec = affinity.pin_rank(local_rank, lw, [r % n_dev for r in range(lw)])
        else:
            torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    from medicaldetectiontoolkit_amd import training
    from medicaldetectiontoolkit_amd.utils import fused_epilogue
    fused_epilogue.ENABLED = bool(args.fused_epilogue)
    fused_epilogue.BWD_DATA_AS_FWD = bool(args.conv_bwd_as_fwd)
    fused_epilogue.STEM_SPACE_TO_DEPTH = bool(args.stem_s2d)
    fused_epilogue.POOL_CHANNELS_LAST = bool(args.pool_cl)
    fused_epilogue.WGRAD_1X1 = bool(args.wgrad_1x1)
    fused_epilogue.UPSAMPLE_CL = bool(args.upsample_cl)
    fused_epilogue.CONV3_SMALL = bool(args.conv3_small)
    fused_epilogue.RES_TAP = bool(args.res_tap)
    fused_epilogue.STRIDE_TAP = bool(args.stride_tap)
    fused_epilogue.UPSAMPLE_NEAREST_CL = bool(args.upsample_nearest_cl)
    fused_epilogue.CONV3_SMALL_EPILOGUE = bool(args.conv3_small_epilogue)
    fused_epilogue.STEM_WGRAD = bool(args.stem_wgrad)
    fused_epilogue.STEM_FWD = bool(args.stem_fwd)
    fused_epilogue.S2D_GENERAL = bool(args.conv_s2d)
    fused_epilogue.S221_FWD = bool(args.conv_s221_fwd)
    fused_epilogue.S221_DGRAD = bool(args.conv_s221_dgrad)
    fused_epilogue.CONV_WIN = bool(args.conv_win)
    fused_epilogue.CONV_WIN_WGRAD = bool(args.conv_win_wgrad)
    from medicaldetectiontoolkit_amd.configs import Configs
    from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
    from medicaldetectiontoolkit_amd.models import mrcnn, retina_unet
    from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device
    _roi_align_impl.ROI_ALIGN_CHANNELS_LAST = bool(args.roialign_cl)
    mrcnn.HEAD_AS_LINEAR = bool(args.head_as_linear)
    mrcnn.MERGE_RPN_HEADS = bool(args.merge_rpn_heads)
    mrcnn.SPARSE_RPN_LOSS = bool(args.sparse_rpn_loss)
    mrcnn.FUSED_GLUE = bool(args.fused_glue)
    mrcnn.RPN_HEADS_FUSED = bool(args.rpn_heads_fused)
    mrcnn.SHARED_PYRAMID_GRAD = bool(args.shared_pyramid_grad)
    fused_epilogue.BIAS_GRAD_IN_LAUNCH = bool(args.bias_grad_in_launch)
    fused_epilogue.FLIP_BATCHED = bool(args.flip_batched)
    fused_epilogue.BIAS_BWD_NO_COPY = bool(args.bias_bwd_no_copy)
    fused_epilogue.CONV1X1_FWD = bool(args.conv1x1_fwd)
    fused_epilogue.SEG_HEAD_COMPOSED = bool(args.seg_head_composed)
    fused_epilogue.CONV_C0 = bool(args.conv_c0)
    fused_epilogue.STEM_POOL_FUSED = bool(args.stem_pool_fused)
    fused_epilogue.CONV1X1_BWD = bool(args.conv1x1_bwd)
    fused_epilogue.BIAS_GRAD_TRANSPOSE = bool(args.bias_grad_transpose)
    fused_epilogue.LATERAL_UPSAMPLE_FUSED = bool(args.lateral_upsample_fused)

    # MIOpen's immediate-mode heuristics pick naive 3D solvers for the 18/36/72-channel convolutions of this
    # backbone (3.2 s per step); the exhaustive find selects im2col+GEMM / CK kernels (42x faster, profiles/).
    torch.backends.cudnn.benchmark = True
    patch = [int(v) for v in args.patch.split(",")]
    cf = Configs(dim=3, model=args.model, patch_size=patch, batch_size=args.batch, channels_last=bool(args.channels_last))
    torch.manual_seed(0)          # identical initial weights on every rank
    net = (mrcnn if args.model == "mrcnn" else retina_unet).net(cf, device=dev)
    torch.manual_seed(1000 + rank)
    sync = training.FlatGradAllReduce(net) if world > 1 else None
    opt = training.build_optimizer(net, cf, fused=bool(args.fused_adam), flat=bool(args.flat_adam), grad_sync=sync)
    # rank-disjoint synthetic patch streams, generated before the timed region (the reference's loader runs in
    # background worker processes and is excluded from its own per-batch timing, exec.py:68-77)
    pool = [make_batch(patch, args.batch, seed=1000 * rank + i) for i in range(3)]
    if not args.host_batches:   # inputs resident in HBM when the timed region starts (measurement contract)
        pool = [to_device(b, dev) for b in pool]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the step.  Headline = the step launched eagerly (--graph 0, default) or as ONE hipGraph replay (--graph 1); the other form is
    # timed as an A/B leg in the same process, so every line carries both.  Same-box pairs (profiles/r04/r04_ab_same_box.txt): eager
    # 190.4 patches/s, graphed 182.5 -- at one rank on a fast host the eager step is already GPU-bound and the replay costs ~1 us of GPU
    # time per graph node with the runtime's packet capture off (medicaldetectiontoolkit_amd/__init__.py); what the graph buys is the HOST:
    # ~4.5 ms instead of 37-40 ms of host work per step.
    graph_rec = {"headline": bool(args.graph)}
    use_graph = bool(args.graph) and args.model == "mrcnn"
    if use_graph and not args.no_graph_preflight:
        ok, note, _ = _graph_preflight(args, local_dev)      # a capture crash must kill a child, never the bench line
        graph_rec["preflight"] = note
        use_graph = ok
    graph_rec["used_for_headline"] = use_graph
    # graphed headline: capture FIRST, before any eager step of this net (graph_preflight_main explains)
    exec_form = args.step_form == "exec"
    # the read-out mode of the headline step: the packed read-out (for the Retina U-Net also the uint8 label map) travels with an asynchronous copy and
    # the entries of step i are consumed while step i + 1 is queued ("deferred")
    mon_mode = False if not exec_form else "deferred"
    if exec_form and args.model == "mrcnn":
        cf.run_detection_mask_head_in_training = True       # mrcnn.py:1046-1048: the reference runs it in every training step
    gstep = training.GraphedTrainStep(net, opt, grad_sync=sync, gmax=args.gmax, monitor=mon_mode, with_masks=exec_form) if use_graph else None
    consumed = [0, None]               # what exec.py:76-79 reads every batch: characters / boxes / values consumed, the last log line

    def run_step(b):
        r = gstep(b) if gstep is not None else training.train_step(net, opt, b, grad_sync=sync, monitor=mon_mode)
        if "logger_string" in r:       # (deferred: absent on the very first call only)
            consumed[0] += len(r["logger_string"]) + len(r["boxes"]) + len(r["monitor_values"])
            consumed[1] = r["logger_string"]
        return r

    def host_work_of(g):
        hm, g.host_ms = g.host_ms, None
        c = max(1, hm.pop("calls", 1))
        hw = {k: round(v / c, 3) for k, v in hm.items()}
        hw["work_total"] = round(sum(v for k, v in hw.items() if k != "ring_wait_backpressure"), 3)
        return hw

    def eager_step(b):                 # the headline step's form, launched eagerly (A/B legs)
        r = training.train_step(net, opt, b, grad_sync=sync, monitor=mon_mode)
        if "logger_string" in r:
            consumed[0] += len(r["logger_string"]) + len(r["boxes"]) + len(r["monitor_values"])
        return r

    coll_cpu = [0.0]                   # issue-thread CPU seconds spent inside the gradient collective (launch + wait): reported apart -- with the gloo debug
    if sync is not None:               # backend that is a CPU reduction of the 20 MB buffer, not a property of the RCCL path
        for name in ("finish", "finish_all"):
            def timed(_orig=getattr(sync, name)):
                c0 = time.thread_time()
                try:
                    return _orig()
                finally:
                    coll_cpu[0] += time.thread_time() - c0
            setattr(sync, name, timed)

    for i in range(max(args.warmup, 1 if use_graph else 0)):
        run_step(pool[i % len(pool)])
    # settle steps (untimed, the same fixed count on every rank): the FIRST process on a fresh box runs its steps 5 .. 25 at 34.9 ms instead of 31.8 (measured:
    # three bench runs in a row on one fresh box gave 229.6 / 249.2 / 251.8 patches/s for --warmup 5, 251.5 for --warmup 25; the later legs of the first run were at
    # full speed) -- box-level warm-up (MIOpen's kernel cache being written, clocks), not part of the step.  K steps are timed exactly, after W + settle untimed ones.
    settle = max(0, 25 - args.warmup) if args.settle < 0 else args.settle
    for i in range(settle):
        run_step(pool[i % len(pool)])
    barrier()
    prof = None
    if gstep is not None:
        gstep.host_ms = {}
    else:
        _roi_align_impl.PROFILE = []          # the RoIAlign backward launches of the timed steps, event-timed (roofline in-step variant)
    counts = []                        # (valid, positive) sampled RoIs of every timed step: device scalars, read after the timed region
    cpu0 = time.process_time()         # CPU seconds of ALL threads of this rank (issue thread, autograd engine, OpenMP / collective helper threads)
    thr0 = time.thread_time()          # CPU seconds of the ISSUE thread alone: the Python + launch work of a step, what must not grow with the rank count
    coll0 = coll_cpu[0]
    t0 = time.time()
    for i in range(args.steps):
        r_i = run_step(pool[i % len(pool)])
        if "sample_counts" in r_i:
            counts.append(r_i["sample_counts"])
    host_issue = time.time() - t0      # the host has launched everything (it runs ahead of the GPU while the step is GPU-bound)
    host_cpu = time.process_time() - cpu0
    host_thr = time.thread_time() - thr0
    host_coll = coll_cpu[0] - coll0
    barrier()
    elapsed = time.time() - t0
    timed_batches = None
    if counts and args.model == "mrcnn":
        cv = [(float(a), float(b)) for a, b in (counts if gstep is None else counts[-1:])]
        timed_batches = {"roi_slots_per_step": int(cf.train_rois_per_image * args.batch),
                         "valid_rois_per_step": round(float(np.mean([c[0] for c in cv])), 2), "positive_rois_per_step": round(float(np.mean([c[1] for c in cv])), 2),
                         "note": "random synthetic GT on random-init weights: few proposals overlap a GT box, so most RoI-head slots are padding (zero-weighted rows "
                                 "that still run through the heads at full fixed size) and the mask / box losses see few or no positives"}
    if gstep is not None:
        # a replay blocks while the previous replay of the same graph is still executing (measured: 40 ms inside hipGraphLaunch with the
        # GPU busy, 3.7-4.9 ms with the GPU idle): host_issue is therefore ~ the GPU time; the host's own WORK is in host_ms_per_step
        graph_rec["host_ms_per_step"] = host_work_of(gstep)
    else:
        prof, _roi_align_impl.PROFILE = _roi_align_impl.PROFILE, None

    # ---- A/B leg: the other form of the step, same net / optimizer / batches, warmed up like the headline
    eager_rec, graphed_rec = None, None
    n_ab = max(2, min(args.steps, 8))
    # ---- round 5's headline form as a named leg: no read-out, no mask head over the detections (what the step costs without exec.py's consumers)
    no_readout_rec = None
    if exec_form:
        prev_mh = getattr(cf, "run_detection_mask_head_in_training", False)
        cf.run_detection_mask_head_in_training = False
        try:
            n_nr = max(2, min(args.steps, 10))
            for i in range(2):
                training.train_step(net, opt, pool[i % len(pool)], grad_sync=sync, monitor=False)
            barrier()
            tn = time.time()
            for i in range(n_nr):
                training.train_step(net, opt, pool[i % len(pool)], grad_sync=sync, monitor=False)
            th_n = time.time() - tn
            barrier()
            tn = time.time() - tn
            no_readout_rec = {"value": round(args.batch * world * n_nr / tn, 3), "unit": "patches/s", "steps": n_nr, "ms_per_step": round(tn / n_nr * 1e3, 2),
                              "host_issue_ms_per_step": round(th_n / n_nr * 1e3, 2),
                              "note": "train_step(monitor=False), eager: forward + backward + Adam with every loss term and gradient, WITHOUT the per-batch read-out of "
                                      "exec.py:76-79 and without the mask head over the detections (round 5's `value` form)"}
        finally:
            cf.run_detection_mask_head_in_training = prev_mh
    if use_graph and not args.no_eager_leg:
        for i in range(3):
            eager_step(pool[i % len(pool)])
        barrier()
        _roi_align_impl.PROFILE = []
        te = time.time()
        for i in range(n_ab):
            eager_step(pool[i % len(pool)])
        th_e = time.time() - te
        barrier()
        te = time.time() - te
        prof, _roi_align_impl.PROFILE = _roi_align_impl.PROFILE, None
        eager_rec = {"value": round(args.batch * world * n_ab / te, 3), "unit": "patches/s", "steps": n_ab, "ms_per_step": round(te / n_ab * 1e3, 2),
                     "host_issue_ms_per_step": round(th_e / n_ab * 1e3, 2),
                     "note": "training.train_step: the same step with every kernel launched eagerly"}
    # ---- the same step with the RPN losses differentiated through the DENSE RPN outputs, as the reference's autograd graph does
    # (mrcnn.py:176-240 on the outputs of :987-1003): identical gradients (tests/test_models_gpu.py::test_sparse_rpn_loss_step_equals_
    # dense_graph_step), 6.6 ms more convolution work on zeros -- reported beside `value` so that the gain of the sampled-anchor form is visible
    dense_rpn_rec = None
    if args.model == "mrcnn" and not use_graph and args.sparse_rpn_loss and not args.no_dense_rpn_leg:
        mrcnn.SPARSE_RPN_LOSS = False
        try:
            for i in range(3):
                eager_step(pool[i % len(pool)])
            barrier()
            td = time.time()
            for i in range(n_ab):
                eager_step(pool[i % len(pool)])
            barrier()
            td = time.time() - td
            dense_rpn_rec = {"value": round(args.batch * world * n_ab / td, 3), "unit": "patches/s", "steps": n_ab, "ms_per_step": round(td / n_ab * 1e3, 2),
                             "note": "same net / optimizer / batches with --sparse-rpn-loss 0: RPN losses differentiated through the dense RPN outputs like the "
                                     "reference's graph (same gradients; the dense graph back-propagates zeros through conv_shared on every level)"}
        finally:
            mrcnn.SPARSE_RPN_LOSS = True
    child_legs = None
    if not use_graph and args.model == "mrcnn" and not args.no_graph_leg and rank == 0 and world == 1:
        ok, note, child_legs = _graph_preflight(args, local_dev, legs=True)      # all graph work of an eager line: in a child process
        graph_rec["legs_child"] = note
        graphed_rec = child_legs.get("graphed_step") if (ok and child_legs) else {"failed": note}

    # ---- the in-step RoIAlign backward with the RoI heads FULL: GT boxes derived from the net's own proposals (as
    # tests/golden/make_step_golden.py does), so that the target layer finds positives and all train_rois_per_image slots are valid
    prof48 = None
    heads_full_rec = None
    if world == 1 and args.model == "mrcnn" and not args.no_roofline:
        try:
            from medicaldetectiontoolkit_amd.utils.synthetic_data import batch_with_gt_from_proposals
            b48 = batch_with_gt_from_proposals(net, cf, pool[0] if not args.host_batches else to_device(pool[0], dev), dev)
            for _ in range(2):
                eager_step(b48)
            torch.cuda.synchronize()
            _roi_align_impl.PROFILE = []
            for _ in range(12):          # 12 launches of the mask head's pyramid backward (4 were too few: 0.598 .. 0.643 between runs)
                eager_step(b48)
            torch.cuda.synchronize()
            prof48, _roi_align_impl.PROFILE = _roi_align_impl.PROFILE, None
            # the same step TIMED on that batch (VERDICT r4 "weak" 2): every RoI-head slot that can be valid is, positives exist, the mask and
            # box losses are live -- no event hooks in these steps
            c48 = []
            barrier()
            t48 = time.time()
            for _ in range(n_ab):
                c48.append(eager_step(b48)["sample_counts"])
            barrier()
            t48 = time.time() - t48
            heads_full_rec = {"value": round(args.batch * n_ab / t48, 3), "unit": "patches/s", "steps": n_ab, "ms_per_step": round(t48 / n_ab * 1e3, 2),
                              "roi_slots_per_step": int(cf.train_rois_per_image * args.batch),
                              "valid_rois_per_step": round(float(np.mean([float(a) for a, _ in c48])), 2),
                              "positive_rois_per_step": round(float(np.mean([float(b) for _, b in c48])), 2),
                              "note": "the headline step on a batch whose GT boxes are two large disjoint proposals of the net's own RPN per element "
                                      "(utils/synthetic_data.batch_with_gt_from_proposals, the construction of tests/golden/make_step_golden.py): positives "
                                      "exist, the RoI heads are full; the weights keep training on this one batch, so the counts drift over the steps"}
        except Exception as e:
            prof48 = None
            graph_rec["instep48_failed"] = repr(e)[:200]
            _roi_align_impl.PROFILE = None

    # the same steps fed from host numpy batches (the reference uploads inside train_forward, mrcnn.py:869): reported
    # beside `value`, never as `value`
    h2d = None
    if world == 1 and not args.host_batches and not args.no_h2d_leg:
        host_pool = [make_batch(patch, args.batch, seed=i) for i in range(2)]
        n_h2d = max(2, min(args.steps, 5))
        run_step(host_pool[0])
        barrier()
        th = time.time()
        for i in range(n_h2d):
            run_step(host_pool[i % 2])
        barrier()
        h2d = {"value": round(args.batch * n_h2d / (time.time() - th), 3), "unit": "patches/s", "steps": n_h2d,
               "note": "same step with the batch handed over as host numpy arrays (image, GT masks uploaded inside the step, on the step's own thread and stream)"}

    # ---- the step as exec.py consumes it: host numpy batches through training.DevicePrefetcher (upload of batch i + 1
    # behind step i), monitor read-out every step (logger_string / boxes / monitor_values: ONE packed device->host copy), the mask head
    # over the detections run like the reference does (mrcnn.py:1046-1048)
    exec_eq = None
    if world == 1 and args.model == "mrcnn" and not args.no_exec_leg:
        try:
            # both forms: with one read-out sync per step the eager host cannot run ahead of the GPU, the replay can
            if use_graph:
                g_eq = exec_equivalent_leg(net, opt, cf, patch, args, dev, True)
                exec_eq = exec_equivalent_leg(net, opt, cf, patch, args, dev, False)
            else:
                exec_eq = exec_equivalent_leg(net, opt, cf, patch, args, dev, False)
                exec_eq["synchronous_readout_form"] = exec_equivalent_leg(net, opt, cf, patch, args, dev, False, deferred=False)
                g_eq = (child_legs or {}).get("exec_equivalent_graphed")
            if g_eq and "value" in g_eq:
                exec_eq = dict(g_eq, eager_form=exec_eq) if g_eq["value"] >= exec_eq["value"] else dict(exec_eq, graphed_form=g_eq)
        except Exception as e:
            exec_eq = {"failed": repr(e)[:300]}

    t = torch.tensor([elapsed, host_issue, host_cpu, host_thr - host_coll, host_coll], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, host_issue_max, host_cpu_max, host_thr_max, host_coll_max = (float(v) for v in t.tolist())
    # post-step parameters: one checksum per rank; identical weights on every rank <=> min == max
    with torch.no_grad():
        csum = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum().reshape(1)
    cmin, cmax = csum.clone(), csum.clone()
    devices = [torch.cuda.get_device_name(dev)]
    if world > 1:
        dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        devices = [None] * world
        dist.all_gather_object(devices, "rank %d: cuda:%d %s" % (rank, local_dev, torch.cuda.get_device_name(dev)))
    dist_rec = {"world": world, "backend": (dist.get_backend() if world > 1 else None), "devices": devices,
                "param_checksum": float(cmin.item()), "params_identical_across_ranks": bool(cmin.item() == cmax.item()),
                "grad_buckets": (len(sync.bucket_range) if sync is not None and sync.flat is not None else None),
                "rank0_core_affinity": affinity_rec,
                # max over ranks, per timed step: wall time until the rank had issued the step; CPU time of its ISSUE thread (Python + launches: the
                # per-rank host work, which must not grow with N -- tests/test_distributed_gpu.py compares world 8 with world 1 on one box); CPU
                # time of ALL its threads (includes OpenMP workers spinning in their pinned slice and, with the gloo debug backend, the collective's
                # CPU reduction: not a property of the RCCL path)
                "host_issue_ms_per_step_max_over_ranks": round(host_issue_max / args.steps * 1e3, 2),
                "host_issue_thread_cpu_ms_per_step_max_over_ranks": round(host_thr_max / args.steps * 1e3, 2),        # (without the collective, next entry)
                "collective_issue_thread_cpu_ms_per_step_max_over_ranks": round(host_coll_max / args.steps * 1e3, 2),
                "host_cpu_ms_per_step_max_over_ranks": round(host_cpu_max / args.steps * 1e3, 2),
                "host_cores": os.cpu_count()}

    if rank == 0:
        roofline = None if args.no_roofline else roialign_bwd_roofline(cf, args.batch, dev, prof, prof48)
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.model == "mrcnn":
            port = None
            try:
                port = cpu_baseline(cf, net.anchors_f64.cpu().numpy())
            except Exception as e:  # the baseline must never take the bench line down
                port = {"value": None, "unit": "patches/s", "cores": int(torch.get_num_threads()), "kind": "port", "sample": "failed: %r" % (e,)}
            try:        # the reference's own step on the host cores (a full batch), the port beside it
                cpu = cpu_baseline_reference(args)
                cpu["port_one_patch"] = port
            except Exception as e:
                cpu = dict(port, reference_step_failed=repr(e)[:300])
        patches = args.batch * world * args.steps
        out = {
            "metric": "3D patches/sec (train), %s %s" % ("^3".join([str(patch[0]), ""]) if len(set(patch)) == 1 else "x".join(map(str, patch)),
                                                       "Mask R-CNN" if args.model == "mrcnn" else "Retina U-Net"), "value": round(patches / elapsed, 3), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "host_issue_ms_per_step": round(host_issue / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (host numpy batches, PCIe inclusive)" if args.host_batches else " (resident in HBM)"),
            "config": {"workload": "LIDC-shape 3D %s, %s fp32 patches, batch %d per GPU, random-init weights, Adam lr 1e-4" % (
                           "Mask R-CNN (3D RoIAlign + 3D NMS)" if args.model == "mrcnn" else "Retina U-Net", "x".join(map(str, patch)), args.batch),
                       "parallelism": "dp%d (one process per GPU, flat-bucket gradient all-reduce over %s)" % (world, "RCCL" if args.backend == "nccl" else args.backend),
                       "global_batch": args.batch * world,
                       "step_form": ("exec.py:68-79 as the reference runs it: results = net.train_forward(batch) -- forward, losses, "
                                     + ("the mask head over the detections (mrcnn.py:1046-1048), the per-batch read-out exec.py:76-79 consumes (logger_string, box lists, "
                                        "monitor_values: ONE packed device->host copy per step, asynchronous, the entries of step i consumed while step i + 1 is queued) -- "
                                        if (args.model == "mrcnn" and args.step_form == "exec") else
                                        ("the per-batch read-out exec.py:76-79 consumes (logger_string, box lists, seg_preds, monitor_values: packed asynchronous copies, consumed one step late) -- "
                                         if args.step_form == "exec" else "WITHOUT the per-batch read-out (--step-form no-readout) -- "))
                                     + "zero_grad, backward, Adam (exec.py:68-74); every loss term and every parameter gradient of the reference step "
                                     "(tests/test_step_parity_gpu.py pins them against the reference at this configuration).  Batches resident in HBM when the timed region "
                                     "starts (measurement contract); the same step fed host numpy batches through training.DevicePrefetcher is `exec_equivalent`"
                                     + (", the step without read-out and detection mask head is `no_readout_step`.  " if args.step_form == "exec" else ".  ")
                                     + ("The timed batches are random-GT batches on random-init weights: see `timed_batches` for how full the RoI heads were, "
                                        "`heads_full_step` for the same step with full RoI heads.  " if args.model == "mrcnn" else "") + "RPN losses back-propagated "
                                     + ("through the sampled anchors only (same gradients as the dense graph, which is timed as dense_rpn_graph_step)"
                                        if (args.model == "mrcnn" and args.sparse_rpn_loss) else "through the dense RPN outputs"))},
            "timed_batches": timed_batches, "heads_full_step": heads_full_rec,
            "readout_consumed": {"items": consumed[0], "last_logger_string": consumed[1]},
            "no_readout_step": no_readout_rec, "graph": graph_rec, "eager_step": eager_rec, "graphed_step": graphed_rec, "dense_rpn_graph_step": dense_rpn_rec, "exec_equivalent": exec_eq,
            "roofline": roofline, "cpu_baseline": cpu, "h2d_inclusive": h2d, "distributed": dist_rec,
        }
        if world == 1 and not args.no_rccl_selftest:
            try:
                out["distributed"]["rccl_world1_selftest"] = rccl_world1_selftest(net, opt, pool[0] if not args.host_batches else to_device(pool[0], dev), dev,
                                                                                   use_graph=use_graph, gmax=args.gmax)
            except Exception as e:   # reported, never fatal for the bench line
                out["distributed"]["rccl_world1_selftest"] = {"failed": repr(e)}
        if world == 1 and not args.no_secondary and args.model == "mrcnn":
            try:
                del net, opt, pool
                torch.cuda.empty_cache()
                out["secondary"] = secondary_configs()
            except Exception as e:
                out["secondary"] = {"failed": repr(e)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
