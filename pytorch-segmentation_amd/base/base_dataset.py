"""BaseDataSet — the reference's dataset base class (base/base_dataset.py:11-136) with the pixel work moved to the MI355X.

Same constructor (`root, split, mean, std, base_size, augment, val, crop_size, scale, flip, rotate, blur, return_id`) and the same
two hooks a concrete dataset implements (`_set_files`, `_load_data(index) -> (image, label, image_id)`); `len()` = number of files.
What differs: `__getitem__` returns the RAW sample — `(image uint8 [H,W,3], label int32 [H,W], image_id)` — and does no cv2 / PIL
arithmetic on the host.  The resize / rotate / crop / flip / blur / ToTensor / Normalize sequence the reference runs per sample in
its DataLoader workers (`_augmentation` :63-120, `_val_augmentation` :40-61, `__getitem__` :125-136) is applied to the whole batch
on the device by `BaseDataLoader` through `dataloaders.gpu_augment.GPUAugment`, configured from the attributes kept here.
"""
import numpy as np


class BaseDataSet:
    def __init__(self, root, split, mean, std, base_size=None, augment=True, val=False, crop_size=321, scale=True, flip=True,
                 rotate=False, blur=False, return_id=False):
        self.root = root
        self.split = split
        self.mean = mean
        self.std = std
        self.augment = augment
        self.crop_size = crop_size
        # (the reference sets these four only when augment is true, :20-25; unset they mean "off")
        self.base_size = base_size if augment else None
        self.scale = bool(scale) if augment else False
        self.flip = bool(flip) if augment else False
        self.rotate = bool(rotate) if augment else False
        self.blur = bool(blur) if augment else False
        self.val = val
        self.files = []
        self._set_files()
        self.return_id = return_id

    def _set_files(self):
        raise NotImplementedError

    def _load_data(self, index):
        raise NotImplementedError

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        image, label, image_id = self._load_data(index)
        image = np.asarray(image)
        if image.dtype != np.uint8:
            image = image.astype(np.uint8)                     # `np.uint8(image)` of the reference's __getitem__ (:133)
        return np.ascontiguousarray(image), np.ascontiguousarray(np.asarray(label, dtype=np.int32)), image_id

    def __repr__(self):
        return "Dataset: %s\n    # data: %d\n    Split: %s\n    Root: %s" % (self.__class__.__name__, len(self), self.split, self.root)
