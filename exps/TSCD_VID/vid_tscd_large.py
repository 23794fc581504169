"""TSCD-L on ImageNet-VID: the settings of the reference's exps/TSCD_VID/vid_tscd_large.py:12-52 -- 30 classes, clips of 4 local +
12 global frames in training and 1 + 31 in evaluation, a lower proposal clamp of 50 and NO upper clamp (every anchor with
obj * cls >= 0.001 is a proposal)."""
import os

from exps.TSCD_OVIS.ovis_tscd_large import Exp as LargeExp


class Exp(LargeExp):
    def __init__(self):
        super().__init__()
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        self.num_classes = 30
        self.lframe, self.gframe, self.lframe_val, self.gframe_val = 4, 12, 1, 31        # vid_tscd_large.py:27-30
        self.minimal_limit, self.maximal_limit = 50, 0                                   # :42 sets only the lower clamp
        self.output_dir = "./VID_TSCD_outputs/TSCD_Large"
        self.tracker_refine = True
        self.seed = 2024
        # tscd_base.py (TSCD_VID) :141-146: lists of frame paths per video + the annotation caches, relative to the working directory
        self.vid_train_path, self.vid_val_path = "./yolox/data/datasets/train_seq.npy", "./yolox/data/datasets/val_seq.npy"
        self.vid_train_anno_cache_path, self.vid_val_anno_cache_path = "./yolox/data/datasets/train_anno_cache.json", "./yolox/data/datasets/val_anno_cache.json"
        self.local_stride, self.traj_linking, self.tnum = 1, False, -1

    def get_data_loader(self, batch_size, is_distributed, no_aug=False, cache_img=False):
        """ImageNet-VID clips through VIDDataset (the reference's TSCD_VID/tscd_base.py:318-378 without the optional mosaic wrapper) when
        the sequence list exists, the synthetic clip source otherwise (smoke runs, benchmarks)."""
        assert batch_size == self.lframe + self.gframe, "one clip per iteration: -b must equal lframe + gframe"
        if not os.path.isfile(self.vid_train_path):
            return self._synthetic(batch_size, self.input_size[0], is_distributed)
        import torch
        from exps.TSCD_OVIS.tscd_base import _Prefetched
        from yolox.data import TrainTransform, TrainTransformGPU
        from yolox.data.datasets import vid
        tf = TrainTransformGPU if (getattr(self, "gpu_preproc", True) and torch.cuda.is_available()) else TrainTransform
        dataset = vid.VIDDataset(file_path=self.vid_train_path, img_size=self.input_size,
                                 preproc=tf(max_labels=50, flip_prob=self.flip_prob, hsv_prob=self.hsv_prob), lframe=self.lframe,
                                 gframe=self.gframe, dataset_pth=self.data_dir, local_stride=self.local_stride, training=True,
                                 cache_file=self.vid_train_anno_cache_path)
        return _Prefetched(vid.ovis_get_trans_loader(batch_size=batch_size, data_num_workers=self.data_num_workers, dataset=dataset,
                                                     seed=self.seed or 0))

    def get_eval_loader(self, batch_size, tnum=None, data_num_workers=8, formal=False):
        assert batch_size == self.lframe_val + self.gframe_val
        if not os.path.isfile(self.vid_val_path):
            return self._synthetic(batch_size, self.test_size[0])
        import torch.distributed as dist
        from yolox.data.data_augment import Vid_Val_Transform
        from yolox.data.datasets import vid
        dataset = vid.VIDDataset(file_path=self.vid_val_path, img_size=self.test_size, preproc=Vid_Val_Transform(), lframe=self.lframe_val,
                                 gframe=self.gframe_val, val=True, dataset_pth=self.data_dir, tnum=self.tnum if tnum is None else tnum,
                                 formal=formal, traj_linking=self.traj_linking, local_stride=self.local_stride, training=False,
                                 cache_file=self.vid_val_anno_cache_path)
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        return vid.ovis_vid_val_loader(batch_size=batch_size, data_num_workers=data_num_workers, dataset=dataset, rank=rank, world_size=world)
