"""Pascal VOC 2012 / SBD loaders with the reference's surface (dataloaders/voc.py:16-98): `VOC(data_dir, batch_size, split, crop_size,
base_size, scale, num_workers, val, shuffle, flip, rotate, blur, augment, val_split, return_id)`, `.MEAN / .STD`,
`.dataset.num_classes / .palette`.  Files are decoded with PIL on host threads into raw uint8 / int32 arrays; every pixel operation
after that runs on the device (base.BaseDataLoader -> dataloaders.gpu_augment.GPUAugment).

Directory layout (the reference's): <data_dir>/VOCdevkit/VOC2012/{JPEGImages, SegmentationClass, ImageSets/Segmentation/<split>.txt};
the `*_aug` splits list "<image path> <label path>" pairs relative to VOC2012.
"""
import os

import numpy as np

from base import BaseDataLoader, BaseDataSet


def get_voc_palette(num_classes):
    """The PASCAL colour map: bit j of the class index feeds bit (7 - j/3) of channel j % 3."""
    pal = [0] * (3 * num_classes)
    for c in range(num_classes):
        lab, shift = c, 7
        while lab:
            for ch in range(3):
                pal[3 * c + ch] |= ((lab >> ch) & 1) << shift
            lab >>= 3
            shift -= 1
    return pal


def _open(path, dtype):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB") if dtype == np.uint8 else im, dtype=dtype)


class VOCDataset(BaseDataSet):
    def __init__(self, **kwargs):
        self.num_classes = 21
        self.palette = get_voc_palette(self.num_classes)
        super().__init__(**kwargs)

    def _set_files(self):
        self.root = os.path.join(self.root, "VOCdevkit/VOC2012")
        self.image_dir = os.path.join(self.root, "JPEGImages")
        self.label_dir = os.path.join(self.root, "SegmentationClass")
        with open(os.path.join(self.root, "ImageSets/Segmentation", self.split + ".txt")) as f:
            self.files = [line.rstrip() for line in f if line.strip()]

    def _load_data(self, index):
        image_id = self.files[index]
        image = _open(os.path.join(self.image_dir, image_id + ".jpg"), np.uint8)
        label = _open(os.path.join(self.label_dir, image_id + ".png"), np.int32)
        return image, label, image_id.split("/")[-1].split(".")[0]


class VOCAugDataset(BaseDataSet):
    def __init__(self, **kwargs):
        self.num_classes = 21
        self.palette = get_voc_palette(self.num_classes)
        super().__init__(**kwargs)

    def _set_files(self):
        self.root = os.path.join(self.root, "VOCdevkit/VOC2012")
        with open(os.path.join(self.root, "ImageSets/Segmentation", self.split + ".txt")) as f:
            pairs = [line.rstrip().split(" ") for line in f if line.strip()]
        self.files, self.labels = [p[0] for p in pairs], [p[1] for p in pairs]

    def _load_data(self, index):
        image = _open(os.path.join(self.root, self.files[index][1:]), np.uint8)
        label = _open(os.path.join(self.root, self.labels[index][1:]), np.int32)
        return image, label, self.files[index].split("/")[-1].split(".")[0]


class VOC(BaseDataLoader):
    def __init__(self, data_dir, batch_size, split, crop_size=None, base_size=None, scale=True, num_workers=1, val=False, shuffle=False,
                 flip=False, rotate=False, blur=False, augment=False, val_split=None, return_id=False, device=None, seed=None, rank=None, world=None):
        self.MEAN = [0.45734706, 0.43338275, 0.40058118]
        self.STD = [0.23965294, 0.23532275, 0.2398498]
        kwargs = dict(root=data_dir, split=split, mean=self.MEAN, std=self.STD, augment=augment, crop_size=crop_size, base_size=base_size,
                      scale=scale, flip=flip, blur=blur, rotate=rotate, return_id=return_id, val=val)
        if split in ("train_aug", "trainval_aug", "val_aug", "test_aug"):
            dataset = VOCAugDataset(**kwargs)
        elif split in ("train", "trainval", "val", "test"):
            dataset = VOCDataset(**kwargs)
        else:
            raise ValueError("Invalid split name %s" % split)
        super().__init__(dataset, batch_size, shuffle, num_workers, val_split or 0.0, device=device, seed=seed, rank=rank, world=world)
