"""Weights & Biases is optional.  The reference logs through ``wandb`` (fumi/main.py:27-32,108-138;
fumi/utils/utils.py:272,412-419).  When the package is absent (as in this image) a local stand-in keeps the same
call surface: metrics go to ``<log_dir>/metrics.jsonl`` and ``run.dir`` is a plain directory for checkpoints."""
import json
import os
import time


class _LocalRun:
    def __init__(self, root, name):
        self.name = name
        self.dir = os.path.join(root, name)
        os.makedirs(self.dir, exist_ok=True)


class _Config(dict):
    def update(self, other=None, **kw):
        if other is not None and not isinstance(other, dict):
            other = vars(other)
        super().update({k: str(v) for k, v in (other or {}).items()}, **kw)


class LocalWandb:
    def __init__(self):
        self.run = None
        self.config = _Config()
        self._fh = None
        self._pending = None
        self._n_runs = 0
        import atexit
        atexit.register(self._flush_at_exit)   # an exception / Ctrl-C before finish() must not lose the last record

    def _flush_at_exit(self):
        try:
            self.finish()
        except Exception:                      # (the device may be gone; the earlier records are on disk already)
            pass

    def init(self, entity=None, project=None, group=None, job_type=None, save_code=False, dir=None, **kw):
        root = dir or os.environ.get("FUMI_LOG_DIR", "./results")
        self._n_runs += 1                   # two runs of one process within a second must not share a checkpoint directory
        name = f"{job_type or 'run'}-{int(time.time())}-{os.getpid()}-{self._n_runs}"
        rank = 0
        try:                                # episode-sharded runs: ONE run directory, named by rank 0; only rank 0 writes metrics
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                rank = dist.get_rank()
                box = [name]
                dist.broadcast_object_list(box, src=0)
                name = box[0]
        except ImportError:                 # pragma: no cover
            pass
        self.run = _LocalRun(os.path.join(root, "runs"), name)
        self._pending = None
        self._fh = open(os.path.join(self.run.dir, "metrics.jsonl"), "a") if rank == 0 else None
        return self.run

    def _write(self, metrics, step):
        rec = {k: (float(v) if hasattr(v, "__float__") else str(v)) for k, v in metrics.items()}
        if step is not None:
            rec["_step"] = int(step)
        self._fh.write(json.dumps(rec) + "\n")
        self._fh.flush()

    def log(self, metrics, step=None):
        """Written one call late: the values of a training step are lazy scalars (fumi_amd/lazy.py) that wait for the GPU when
        first read -- converting them right away would make every step host-synchronous.  By the next call the step they
        belong to has long finished; ``finish`` writes the last record."""
        if self._fh is None:
            return
        prev, self._pending = self._pending, (dict(metrics), step)
        if prev is not None:
            self._write(*prev)

    def watch(self, *a, **k):
        pass

    def save(self, *a, **k):
        pass

    def restore(self, name, run_path=None, root=None):
        raise FileNotFoundError("wandb is not installed: pass --checkpoint <path/to/best.pth.tar> instead of a run id")

    def finish(self):
        if self._fh is not None:
            if self._pending is not None:
                self._write(*self._pending)
                self._pending = None
            self._fh.close()
            self._fh = None


try:                                   # pragma: no cover - depends on the image
    import wandb as _wandb
    wandb = _wandb
except Exception:                      # ModuleNotFoundError in this image
    wandb = LocalWandb()
