"""CLI driver -- drop-in for ``python fumi/main.py <flags>`` (fumi/main.py:19-156).

    python -m fumi_amd.main --model fumi --dataset synthetic --dropout 0 --batch_size 32 ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m fumi_amd.main ...   (episode-sharded)

Same flags (fumi_amd/utils/utils.py), same flag validation and error types, same train -> test flow and logged
metrics.  ``--dataset inat-anim`` reads the reference's files (inat_anim.json + image_embeddings_<model>.hdf5 / .npy) into
HBM-resident tables (fumi_amd/dataset/inat_anim.py) and raises FileNotFoundError when they are absent; ``--dataset
synthetic`` / ``synthetic-resident`` have the same batch contract and need no files."""
import os
import random
import sys

import numpy as np
import torch

from . import dist as fdist
from . import engine as _engine
from . import hip
from .models import am3, clip, fumi, maml
from .utils import utils
from .utils.wandb_compat import wandb


def get_dataset(args):
    if args.model == "clip" and args.dataset == "synthetic":        # the CLIP baseline consumes supervised mini-batches
        from .dataset.synthetic import get_synthetic_supervised
        return get_synthetic_supervised(args)
    if args.dataset == "supervised-inat-anim":                       # data.py:54-70
        from .dataset.inat_anim import get_supervised_inat_anim
        return get_supervised_inat_anim(args)
    if args.dataset == "synthetic":
        from .dataset.synthetic import get_synthetic
        return get_synthetic(args)
    if args.dataset == "synthetic-resident":       # same task family, drawn from an HBM-resident table by the GPU sampler
        from .dataset.synthetic import get_synthetic_resident
        return get_synthetic_resident(args)
    if args.dataset == "inat-anim":                # the reference's files (fumi/dataset/data.py) -> HBM-resident tables
        from .dataset.inat_anim import get_inat_anim
        return get_inat_anim(args)
    raise NotImplementedError()                    # data.py:73-74 ('cub' needs a torchmeta download, 'supervised-inat-anim' is CLIP's)


# embedding width each image model produces (fumi/main.py:34-44: the three ValueErrors, same messages)
_EMBEDDING_DIMS = {"resnet-152": ("Resnet-152", 2048), "resnet-34": ("Resnet-34", 512)}
# names the test metrics are logged and printed under, in the order the test loops return them
_TEST_METRICS = {"maml": ("loss", "acc"), "fumi": ("loss", "acc"), "am3": ("loss", "acc", "f1", "prec", "rec", "avg_lamda")}


def _check_embedding_flags(args):
    if args.image_embedding_model not in _EMBEDDING_DIMS:
        raise ValueError("Image embedding model must be one of " + " ".join(_EMBEDDING_DIMS))
    pretty, dim = _EMBEDDING_DIMS[args.image_embedding_model]
    if args.im_emb_dim != dim:
        raise ValueError(f"{pretty} outputs {dim}-dimensional embeddings, hence --im_emb_dim should be set to {dim}")


def _restore(args, model, optimizer):
    """--checkpoint: a file, or (like the reference, main.py:61-76) a W&B run id whose best.pth.tar is fetched first."""
    ckpt = args.checkpoint
    if not os.path.exists(ckpt):
        model_path = f"./checkpoints/{args.model}/{args.checkpoint}"
        os.makedirs(model_path, exist_ok=True)
        ckpt = wandb.restore("best.pth.tar", run_path=f"multimodal-image-cls/{args.model}/{args.checkpoint}",
                             root=model_path).name
    opt = optimizer[0] if type(optimizer) == tuple else optimizer
    return utils.load_checkpoint(model, opt, args.device, ckpt)[0]


def main(args):
    check_supported(args)
    family = args.model if args.model in ("maml", "fumi", "clip") else "am3"      # unknown names are AM3, like utils.init_model
    mod = {"maml": maml, "fumi": fumi, "am3": am3, "clip": clip}[family]
    results_path = f"{args.log_dir}/results"
    os.makedirs(results_path, exist_ok=True)
    os.environ["FUMI_LOG_DIR"] = args.log_dir        # where the local W&B stand-in keeps run directories
    os.environ["WANDB_MODE"] = "offline" if args.wandb_offline else "online"
    wandb.init(entity=args.wandb_entity, project=args.wandb_project, group=args.wandb_experiment,
               job_type="eval" if args.evaluate else "train", save_code=True)
    wandb.config.update(args)
    _check_embedding_flags(args)

    train_loader, val_loader, test_loader, dictionary = get_dataset(args)
    max_test_batches = int(args.num_ep_test / args.batch_size)
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(args.seed)

    model = utils.init_model(args, dictionary)
    print(model)
    optimizer = utils.init_optim(args, model)
    if args.checkpoint:
        model = _restore(args, model, optimizer)
    if family == "clip":                                               # supervised baseline (main.py:86-91,109-111)
        opt = optimizer[0] if type(optimizer) == tuple else optimizer
        if not args.evaluate:
            model = clip.training_run(args, model, opt, train_loader, val_loader, n_epochs=args.epochs)
        test_acc = clip.evaluate(args, model, test_loader)
        print(f"\n TEST: \ntest acc: {test_acc}")
        wandb.log({"test/acc": test_acc})
        wandb.finish()
        return dict(test_acc=float(test_acc))
    if not args.evaluate:
        model = mod.training_run(args, model, optimizer, train_loader, val_loader, max_test_batches // 2)

    out = mod.test_loop(args, model, test_loader, max_test_batches)     # (ends with the device status check)
    names = _TEST_METRICS[family]
    values = dict(zip(names, out))
    print("\n TEST: \n" + ", ".join(f"test {k.replace('_', ' ')}: {v}" for k, v in values.items()))
    wandb.log({f"test/{k}": v for k, v in values.items()})
    result = dict(test_loss=float(values["loss"]), test_acc=float(values["acc"]))
    if family == "am3":
        result["test_f1"] = float(values["f1"])
        test_preds, test_true, query_idx, support_idx, support_lamda = out[len(names):len(names) + 5]
        if fdist.world()[0] == 0:                                        # per-query records (main.py:126-138)
            import pandas as pd
            pd.DataFrame({"support_idx": support_idx, "support_lamda": support_lamda, "query_idx": query_idx,
                          "query_preds": test_preds, "query_targets": test_true}
                         ).to_csv(path_or_buf=f"{results_path}/run_{wandb.run.name}.csv")
    wandb.finish()
    return result


def parse_args(argv=None):
    """Flags -> args (+ args.device, fumi/main.py:141-149).  Pure: nothing here touches the engine, so host-only users of the parser
    (tools, tests, a dry run on a login node) get the same namespace the reference builds."""
    args = utils.parser().parse_args(sys.argv[1:] if argv is None else argv)
    use_gpu = (not args.disable_cuda) and torch.cuda.is_available()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    print(f"running on device {args.device}")
    return args


def check_supported(args):
    """What the reference runs and this engine does not, refused where the engine is first needed (the top of ``main``) instead
    of from inside the first meta-step -- with the reference's exception types where it has one."""
    eng = _engine.get_engine()                  # (raises FumiHipError when the shared object has not been built)
    if args.device.type != "cuda" and not getattr(eng, "runs_on_host", False):
        # the reference falls back to the CPU here (fumi/main.py:145-146); this engine is MI355X-only
        raise hip.FumiHipError(
            ("--disable_cuda was given" if args.disable_cuda else "no GPU is visible (torch.cuda.is_available() is False)")
            + ": fumi_amd has no CPU execution path -- every step runs on the MI355X library "
              "(fumi_amd/lib/libfumi_hip.so).  Run the reference itself for a CPU run.")
    family = args.model if args.model in ("maml", "fumi", "clip") else "am3"      # unknown names are AM3, like utils.init_model
    # (--fine_tune with --text_encoder RNN / RNNhid trains the bi-LSTM like the reference, fumi/models/fumi.py:65-67 / am3.py:74-76:
    # every meta-step hands back the adjoint of its text input, csrc/textenc.hip runs the LSTM's backward)
    if family == "am3" and args.text_encoder == "rand" and args.dropout > 0 and not args.evaluate:
        # fumi/models/am3.py:118-126 applies dropout inside h only; the engine's AM3 step draws the masks of g and h together and
        # `rand` replaces g by an identity, so training this combination needs --dropout 0 (the CLI default is 0.25)
        raise NotImplementedError("--model am3 --text_encoder rand trains only with --dropout 0 on this engine "
                                  "(the step's dropout would also hit the identity that stands in for g)")


def _maybe_init_distributed(args):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.device.type == "cuda":
            torch.cuda.set_device(args.device)
            torch.distributed.init_process_group("nccl", device_id=args.device)
        else:
            torch.distributed.init_process_group("gloo")


if __name__ == "__main__":
    _args = parse_args()
    _maybe_init_distributed(_args)
    main(_args)
