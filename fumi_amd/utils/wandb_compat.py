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
        self._n_runs = 0

    def init(self, entity=None, project=None, group=None, job_type=None, save_code=False, dir=None, **kw):
        root = dir or os.environ.get("FUMI_LOG_DIR", "./results")
        self._n_runs += 1                   # two runs of one process within a second must not share a checkpoint directory
        self.run = _LocalRun(os.path.join(root, "runs"), f"{job_type or 'run'}-{int(time.time())}-{os.getpid()}-{self._n_runs}")
        self._fh = open(os.path.join(self.run.dir, "metrics.jsonl"), "a")
        return self.run

    def log(self, metrics, step=None):
        if self._fh is None:
            return
        rec = {k: (float(v) if hasattr(v, "__float__") else str(v)) for k, v in metrics.items()}
        if step is not None:
            rec["_step"] = int(step)
        self._fh.write(json.dumps(rec) + "\n")
        self._fh.flush()

    def watch(self, *a, **k):
        pass

    def save(self, *a, **k):
        pass

    def restore(self, name, run_path=None, root=None):
        raise FileNotFoundError("wandb is not installed: pass --checkpoint <path/to/best.pth.tar> instead of a run id")

    def finish(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


try:                                   # pragma: no cover - depends on the image
    import wandb as _wandb
    wandb = _wandb
except Exception:                      # ModuleNotFoundError in this image
    wandb = LocalWandb()
