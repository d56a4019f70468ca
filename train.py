"""Thin, DDP-aware training driver = the reference's exec.py train() loop (exec.py:30-110) on synthetic patches.

  python train.py --model mrcnn --epochs 2 --batches 20 --exp-dir /tmp/mdt_exp [--resume]
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...

Per epoch: cf.num_train_batches steps (forward, backward, flat-bucket gradient all-reduce, Adam), the per-batch log line of
exec.py:75-79 EVERY batch (monitoring read-out = one packed device->host copy), then a reference-format checkpoint (rank 0).
The batch stream (host numpy dicts, what the reference's batch generator delivers) goes through training.DevicePrefetcher: batch
i + 1 is staged into pinned memory and uploaded on a side stream while step i runs.  --graph 1 replays the device half of the Mask
R-CNN step as one hipGraph (training.GraphedTrainStep: ~4.5 ms instead of ~38 ms of host work per step; the eager step is ~4 %
faster at one rank on a fast host).  Real-data loaders / validation /
model selection are out of scope; this exists so the hot path can be exercised as a training job.
"""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import medicaldetectiontoolkit_amd  # noqa: E402
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
if any(a in ("--graph=1",) for a in sys.argv) or any(a == "--graph" and b == "1" for a, b in zip(sys.argv, sys.argv[1:])):
    # a graphed step needs the runtime's graph packet capture off, decided BEFORE torch is imported (the package import sets nothing)
    medicaldetectiontoolkit_amd.graph_env_setup()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mrcnn", choices=["mrcnn", "retina_unet", "retina_net"])
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--patch", default="128,128,128")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batches", type=int, default=10, help="cf.num_train_batches")
    ap.add_argument("--exp-dir", default="/tmp/mdt_exp")
    ap.add_argument("--resume", action="store_true", help="exec.py --resume_to_checkpoint: continue from <exp-dir>/fold_0/last_checkpoint")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--graph", type=int, default=0, help="1: the device half of the Mask R-CNN step as one hipGraph replay (training.GraphedTrainStep)")
    ap.add_argument("--gmax", type=int, default=8, help="GT objects per batch element the graphed step's fixed-size table holds")
    ap.add_argument("--flat-adam", type=int, default=1, help="1 (default): training.FlatAdam (one C call over flat buffers, torch.optim.Adam's per-parameter "
                    "semantics and state-dict format); 0: torch.optim.Adam itself")
    ap.add_argument("--adam-absent-grad", default="skip", choices=["skip", "zero_after_first"], help="a parameter without a gradient in a step: skip it "
                    "(torch >= 2, default) or update it with a zero gradient once it has had one (torch 0.4.1, the reference's pinned version)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        from medicaldetectiontoolkit_amd.utils import affinity
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        affinity.pin_rank(local_rank, lw, [r % torch.cuda.device_count() for r in range(lw)])     # own cores, near the rank's GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.backends.cudnn.benchmark = True

    from medicaldetectiontoolkit_amd import training
    from medicaldetectiontoolkit_amd.configs import Configs
    from medicaldetectiontoolkit_amd.models import mrcnn, retina_unet
    from medicaldetectiontoolkit_amd.utils import exp_utils
    from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device

    patch = [int(v) for v in args.patch.split(",")]
    cf = Configs(dim=args.dim, model=args.model, patch_size=patch, batch_size=args.batch, channels_last=True,
                 num_epochs=args.epochs, num_train_batches=args.batches)
    torch.manual_seed(0)
    net = (mrcnn if args.model == "mrcnn" else retina_unet).net(cf, device=dev)
    sync = training.FlatGradAllReduce(net) if world > 1 else None
    use_graph = bool(args.graph) and args.model == "mrcnn"
    cf.adam_absent_grad = args.adam_absent_grad
    opt = training.build_optimizer(net, cf, flat=bool(args.flat_adam), grad_sync=sync)
    fold_dir = os.path.join(args.exp_dir, "fold_0")
    start_epoch = 1
    loaded_metrics = None
    if args.resume and os.path.exists(os.path.join(fold_dir, "last_checkpoint", "params.pth")):
        start_epoch, loaded_metrics = exp_utils.load_checkpoint(os.path.join(fold_dir, "last_checkpoint"), net, opt, map_location=dev)
        if rank == 0:
            print("resumed to checkpoint at epoch {}".format(start_epoch), flush=True)
    torch.manual_seed(1000 + rank)

    metrics = loaded_metrics if loaded_metrics else {"train": {"loss": [None]}}     # exec.py continues the loaded dict
    gstep = None
    for epoch in range(start_epoch, cf.num_epochs + 1):
        for g in opt.param_groups:                       # exec.py:59-60: per-epoch learning-rate list
            g["lr"] = cf.learning_rate[min(epoch - 1, len(cf.learning_rate) - 1)]
        t_epoch = time.time()
        losses = []
        if use_graph and gstep is None:
            gstep = training.GraphedTrainStep(net, opt, grad_sync=sync, gmax=args.gmax, monitor="deferred")
        stream = (make_batch(patch, args.batch, seed=((epoch * 100003 + b) * world + rank)) for b in range(cf.num_train_batches))
        t0 = time.time()
        # the per-batch log line of exec.py:75-79, one batch late for Mask R-CNN: the read-out of step i is an asynchronous copy that is
        # consumed after step i + 1 was queued (monitor="deferred"), so neither the host nor the GPU waits for the other
        mode = "deferred" if args.model == "mrcnn" else True

        def log(res, bix):
            losses.append(res["monitor_values"]["loss"])
            if rank == 0:
                print("tr. batch {0}/{1} (ep. {2}) tot {3:.3f}s || {4}".format(
                    bix + 1, cf.num_train_batches, epoch, time.time() - t0, res["logger_string"][:110]), flush=True)
        for bix, batch in enumerate(training.DevicePrefetcher(stream, dev)):
            if gstep is not None:
                res = gstep(batch)
            else:
                res = training.train_step(net, opt, batch, grad_sync=sync, monitor=mode)
            if "logger_string" in res:
                log(res, bix - 1 if res.get("monitor_of_previous_step") else bix)
            t0 = time.time()
        if mode == "deferred":
            last = training.flush_deferred_monitor(gstep if gstep is not None else net)
            if last is not None:
                log(last, cf.num_train_batches - 1)
        metrics["train"]["loss"].append(sum(losses) / max(len(losses), 1))
        exp_utils.save_last_checkpoint(fold_dir, net, opt, epoch, metrics)
        if rank == 0:
            print("epoch {0} done in {1:.1f}s, mean loss {2:.4f}, {3:.1f} patches/s".format(
                epoch, time.time() - t_epoch, metrics["train"]["loss"][-1],
                cf.num_train_batches * args.batch * world / (time.time() - t_epoch)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
