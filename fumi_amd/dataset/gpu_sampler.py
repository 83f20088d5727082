"""GPU-resident episode sampler.

The reference builds every task on the training thread: torchmeta picks N classes, ``InatAnim.__getitem__`` reads ALL image
embeddings of each class from HDF5 (fumi/dataset/data.py:533-549), ``ClassSplitter`` shuffles and cuts K / Q samples, and the
collate stacks B tasks and copies them to the device (``num_workers 0``, fumi/utils/utils.py:95-98).  Once the meta-step
takes 0.35 ms that loader is the job.  Here the embedding table ([n_images, D] fp32; iNat-Anim is ~1.6 GB of the 288 GB of
HBM), the per-class image lists (CSR) and the per-class text rows stay on the device; a meta-batch is two launches
(``fumi_hip_sample_episodes`` + row gathers) and has the loader's batch contract:

    batch = {'train': ([idx i64 [B,S], text [B,S,Dt] f32 | [B,S,L] i64, im f32 [B,S,D]], targets i64 [B,S]),
             'test' : same with Qn = N * num_shots_test rows}

with the class-major order / categorical labels of torchmeta's ConcatTask.  Sampling is reproducible from (seed, step).
``zero_copy=True`` hands the FuMI engine ``hip.RowRef``s (table + indices) in place of the gathered image rows: the X-panel
kernels then read the rows where they lie in the table (fumi_hip_fumi_step_indexed)."""
import numpy as np
import torch

from .. import hip


class GpuEpisodeSampler:
    def __init__(self, images, class_of_image, class_text, num_ways, num_shots, num_shots_test, batch_size, seed=123,
                 length=None, zero_copy=False, row_ids=None, skip_small_classes=None, torchmeta_tasks=False):
        """images [n_images, D] fp32 (moved to the device once), class_of_image [n_images] ints (category of every row, as
        inat_anim.json's annotations give it), class_text [C, Dt] fp32 or [C, L] int64 tokens (one row per class: the text of
        a sample is its class description, data.py:543-549).  row_ids [n_images] ints: the id reported for every table row in
        the batch's index field (the dataset's image ids, data.py:568-571); default: the row number.
        A class with fewer than num_shots + num_shots_test images cannot fill an episode: torchmeta's ClassSplitter raises
        ValueError only when such a class is drawn, so a dataset the reference accepts must still load: by default (None) the
        sampler warns and samples among the classes that are large enough (the kernel never sees an under-populated class:
        no wrapped indices, no support rows leaking into the query set); ``True`` does the same silently, ``False`` is
        strict and raises up front.
        torchmeta_tasks=True reproduces torchmeta's task semantics: the class slots of a task get a
        random permutation of the labels 0..N-1 (Categorical) and a class tuple drawn again has the same support / query
        members (ClassSplitter seeds its shuffle with hash(task) + seed); False keeps label n for slot n."""
        coi = np.asarray(class_of_image, dtype=np.int64)
        C = int(class_text.shape[0])
        if coi.min() < 0 or coi.max() >= C or len(coi) != images.shape[0]:
            raise ValueError("class_of_image must hold one category in [0, C) per image row")
        self.N, self.K, self.Q = int(num_ways), int(num_shots), int(num_shots_test)
        counts = np.bincount(coi, minlength=C)
        small = np.flatnonzero(counts < self.K + self.Q)
        self.class_ids = None                      # sampled class slot -> row of class_text (None: identity)
        if small.size:
            if skip_small_classes is None:
                import warnings
                warnings.warn(f"{small.size} of {C} classes have fewer than num_shots + num_shots_test = {self.K + self.Q} images "
                              f"and are never sampled (torchmeta's ClassSplitter raises when it draws one)")
            elif not skip_small_classes:
                raise ValueError(f"{small.size} of {C} classes (first: {int(small[0])} with {int(counts[small[0]])} images) have "
                                 f"fewer than num_shots + num_shots_test = {self.K + self.Q} images; torchmeta's ClassSplitter "
                                 f"raises for such a class -- pass skip_small_classes=True to sample among the others")
            keep = np.flatnonzero(counts >= self.K + self.Q)
            if keep.size < self.N:
                raise ValueError("fewer than num_ways classes have num_shots + num_shots_test images")
            self.class_ids = torch.from_numpy(keep.astype(np.int64))
            rows = np.flatnonzero(np.isin(coi, keep))                      # table rows of the eligible classes, file order
            order = rows[np.argsort(coi[rows], kind="stable")]
            counts = counts[keep]
        else:
            order = np.argsort(coi, kind="stable")
        self.C, self.B = len(counts), int(batch_size)
        if self.C < self.N:
            raise ValueError("fewer classes than num_ways")
        self.class_ptr_host = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.class_items_host = order.astype(np.int64)
        self.dev = images.device if images.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self.images = images.to(self.dev, torch.float32).contiguous()
        self.class_ptr = torch.from_numpy(self.class_ptr_host).to(self.dev)
        self.class_items = torch.from_numpy(self.class_items_host).to(self.dev)
        self.class_text = class_text.to(self.dev).contiguous()
        if self.class_ids is not None:
            self.class_ids = self.class_ids.to(self.dev)
        self.seed, self.length = int(seed), length
        self.torchmeta_tasks = bool(torchmeta_tasks)
        self.zero_copy = bool(zero_copy)       # hand out RowRefs into the table instead of gathered image rows (FuMI engine only)
        S, Qn = self.N * self.K, self.N * self.Q
        lab = torch.arange(self.N, device=self.dev, dtype=torch.int64)
        self.y_s = lab.repeat_interleave(self.K).unsqueeze(0).expand(self.B, S).contiguous()      # ConcatTask order
        self.y_q = lab.repeat_interleave(self.Q).unsqueeze(0).expand(self.B, Qn).contiguous()
        self.row_ids = None if row_ids is None else torch.as_tensor(np.asarray(row_ids, dtype=np.int64)).to(self.dev)
        if self.row_ids is not None and self.row_ids.numel() != self.images.shape[0]:
            raise ValueError("row_ids must hold one id per image row")
        self.ws = hip.Workspace.get(self.dev)

    def batch(self, step):
        B, N, K, Q = self.B, self.N, self.K, self.Q
        y_s, y_q = self.y_s, self.y_q
        if self.torchmeta_tasks:
            cls, lab, it_s, it_q = hip.sample_episodes_tm(self.ws, self.seed, step, B, N, K, Q, self.class_ptr, self.class_items, True)
            y_s, y_q = lab.repeat_interleave(K, dim=1), lab.repeat_interleave(Q, dim=1)
        else:
            cls, it_s, it_q = hip.sample_episodes(self.ws, self.seed, step, B, N, K, Q, self.class_ptr, self.class_items)
        if self.zero_copy:
            x_s, x_q = hip.RowRef(self.images, it_s.view(B, N * K)), hip.RowRef(self.images, it_q.view(B, N * Q))
        else:
            x_s = hip.gather_rows(self.ws, self.images, it_s.view(-1)).view(B, N * K, -1)
            x_q = hip.gather_rows(self.ws, self.images, it_q.view(-1)).view(B, N * Q, -1)
        if self.class_ids is not None:
            cls = self.class_ids[cls]
        t_cls = hip.gather_rows(self.ws, self.class_text, cls.view(-1)).view(B, N, -1)           # one text row per class slot
        text_s = t_cls.repeat_interleave(K, dim=1)
        text_q = t_cls.repeat_interleave(Q, dim=1)
        id_s, id_q = it_s.view(B, N * K), it_q.view(B, N * Q)
        if self.row_ids is not None:
            id_s, id_q = self.row_ids[id_s], self.row_ids[id_q]
        return {'train': ([id_s, text_s, x_s], y_s), 'test': ([id_q, text_q, x_q], y_q)}

    def __iter__(self):
        i = 0
        while self.length is None or i < self.length:
            yield self.batch(i)
            i += 1
