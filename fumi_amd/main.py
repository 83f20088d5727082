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
from .models import am3, fumi, maml
from .utils import utils
from .utils.wandb_compat import wandb


def get_dataset(args):
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


def main(args):
    results_path = f"{args.log_dir}/results"
    os.makedirs(results_path, exist_ok=True)
    os.environ["FUMI_LOG_DIR"] = args.log_dir        # where the local W&B stand-in keeps run directories
    job_type = "eval" if args.evaluate else "train"
    os.environ['WANDB_MODE'] = 'offline' if args.wandb_offline else 'online'
    wandb.init(entity=args.wandb_entity, project=args.wandb_project, group=args.wandb_experiment, job_type=job_type,
               save_code=True)
    wandb.config.update(args)

    if args.image_embedding_model not in ["resnet-152", "resnet-34"]:
        raise ValueError("Image embedding model must be one of resnet-152 resnet-34")
    if args.image_embedding_model == "resnet-152" and args.im_emb_dim != 2048:
        raise ValueError("Resnet-152 outputs 2048-dimensional embeddings, hence --im_emb_dim should be set to 2048")
    if args.image_embedding_model == "resnet-34" and args.im_emb_dim != 512:
        raise ValueError("Resnet-34 outputs 512-dimensional embeddings, hence --im_emb_dim should be set to 512")

    train_loader, val_loader, test_loader, dictionary = get_dataset(args)
    max_test_batches = int(args.num_ep_test / args.batch_size)

    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)

    model = utils.init_model(args, dictionary)
    print(model)
    optimizer = utils.init_optim(args, model)

    if args.checkpoint:
        ckpt = args.checkpoint
        if not os.path.exists(ckpt):                         # a W&B run id, like the reference (main.py:61-76)
            model_path = f"./checkpoints/{args.model}/{args.checkpoint}"
            os.makedirs(model_path, exist_ok=True)
            ckpt = wandb.restore("best.pth.tar", run_path=f"multimodal-image-cls/{args.model}/{args.checkpoint}",
                                 root=model_path).name
        opt = optimizer[0] if type(optimizer) == tuple else optimizer
        model, _ = utils.load_checkpoint(model, opt, args.device, ckpt)

    mod = {"maml": maml, "fumi": fumi}.get(args.model, am3)
    if not args.evaluate:
        model = mod.training_run(args, model, optimizer, train_loader, val_loader, max_test_batches // 2)

    if args.model == "maml":
        test_loss, test_acc = maml.test_loop(args, model, test_loader, max_test_batches)
    elif args.model == "fumi":
        test_loss, test_acc, _, _ = fumi.test_loop(args, model, test_loader, max_test_batches)
    if args.model in ("maml", "fumi"):
        print(f"\n TEST: \ntest loss: {test_loss}, test acc: {test_acc}")
        wandb.log({"test/acc": test_acc, "test/loss": test_loss})
        result = dict(test_loss=float(test_loss), test_acc=float(test_acc))
    else:
        (test_loss, test_acc, test_f1, test_prec, test_rec, test_avg_lamda, test_preds, test_true, query_idx,
         support_idx, support_lamda) = am3.test_loop(args, model, test_loader, max_test_batches)
        print(f"\n TEST: \ntest loss: {test_loss}, test acc: {test_acc},\ntest f1: {test_f1}, test prec: {test_prec}, "
              f"test rec: {test_rec}, test avg lamda: {test_avg_lamda}")
        wandb.log({"test/acc": test_acc, "test/f1": test_f1, "test/prec": test_prec, "test/rec": test_rec,
                   "test/loss": test_loss, "test/avg_lamda": test_avg_lamda})
        if fdist.world()[0] == 0:
            import pandas as pd
            pd.DataFrame({"support_idx": support_idx, "support_lamda": support_lamda, "query_idx": query_idx,
                          "query_preds": test_preds, "query_targets": test_true}
                         ).to_csv(path_or_buf=f"{results_path}/run_{wandb.run.name}.csv")
        result = dict(test_loss=float(test_loss), test_acc=float(test_acc), test_f1=float(test_f1))
    wandb.finish()
    return result


def parse_args(argv=None):
    args = utils.parser().parse_args(sys.argv[1:] if argv is None else argv)
    use_gpu = (not args.disable_cuda) and torch.cuda.is_available()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    print(f"running on device {args.device}")
    return args


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
