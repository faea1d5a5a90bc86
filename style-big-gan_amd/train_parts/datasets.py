"""Image datasets for real-data runs (host side only -- the metric uses synthetic reals).

Same surface as the reference's ``train_parts/datasets.py`` (:27-248): a ``Dataset`` base class yielding ``(uint8 CHW image, float32
label)`` with ``max_size`` sub-sampling, x-flip doubling, lazily loaded labels and the ``image_shape / num_channels / resolution /
label_shape / label_dim / has_labels / has_onehot_labels / get_label / get_details`` accessors the trainer and the snapshot grid read, and
``ImageFolderDataset`` (registry name ``image_folder``) over a directory tree or a zip archive of images with an optional
``dataset.json`` (``{"labels": [[relative file name, class index | vector], ...]}``).  Images stay uint8 until they are on the
device (one byte per value over PCIe; ``x / 127.5 - 1`` happens there, reference trainers.py:716).
"""
import json
import os
import zipfile

import numpy as np
import PIL.Image
import torch

from .. import utils
from ..utils import EasyDict

datasets = utils.ClassRegistry()


class Dataset(torch.utils.data.Dataset):
    def __init__(self, name, raw_shape, max_size=None, use_labels=False, xflip=False, random_seed=0):
        self._name = name
        self._raw_shape = list(raw_shape)                   # [count, C, H, W] of the stored images
        self._use_labels = use_labels
        self._raw_labels = None
        self._label_shape = None
        order = np.arange(self._raw_shape[0], dtype=np.int64)
        if max_size is not None and order.size > max_size:  # a fixed random subset, kept in storage order
            np.random.RandomState(random_seed).shuffle(order)
            order = np.sort(order[:max_size])
        flips = np.zeros(order.size, dtype=np.uint8)
        if xflip:                                           # every image once as stored, once mirrored
            order, flips = np.tile(order, 2), np.concatenate([flips, np.ones_like(flips)])
        self._raw_idx, self._xflip = order, flips

    # -- to be provided by subclasses
    def _load_raw_image(self, raw_idx):
        raise NotImplementedError

    def _load_raw_labels(self):
        raise NotImplementedError

    def close(self):
        pass

    # -- labels
    def _get_raw_labels(self):
        if self._raw_labels is None:
            labels = self._load_raw_labels() if self._use_labels else None
            if labels is None:
                labels = np.zeros([self._raw_shape[0], 0], dtype=np.float32)
            assert isinstance(labels, np.ndarray) and labels.shape[0] == self._raw_shape[0] and labels.dtype in (np.float32, np.int64)
            if labels.dtype == np.int64:
                assert labels.ndim == 1 and np.all(labels >= 0)
            self._raw_labels = labels
        return self._raw_labels

    def get_label(self, idx):
        label = self._get_raw_labels()[self._raw_idx[idx]]
        if label.dtype == np.int64:                         # class index -> one-hot
            onehot = np.zeros(self.label_shape, dtype=np.float32)
            onehot[label] = 1
            return onehot
        return label.copy()

    def get_details(self, idx):
        raw = int(self._raw_idx[idx])
        return EasyDict(raw_idx=raw, xflip=bool(self._xflip[idx]), raw_label=self._get_raw_labels()[raw].copy())

    # -- torch Dataset protocol
    def __len__(self):
        return self._raw_idx.size

    def __getitem__(self, idx):
        image = self._load_raw_image(self._raw_idx[idx])
        assert isinstance(image, np.ndarray) and image.dtype == np.uint8 and list(image.shape) == self.image_shape
        if self._xflip[idx]:
            image = image[:, :, ::-1]
        return image.copy(), self.get_label(idx)

    def __getstate__(self):                                 # worker processes reload the labels themselves
        return dict(self.__dict__, _raw_labels=None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties
    name = property(lambda self: self._name)
    image_shape = property(lambda self: list(self._raw_shape[1:]))
    num_channels = property(lambda self: self.image_shape[0])
    has_labels = property(lambda self: any(x != 0 for x in self.label_shape))
    has_onehot_labels = property(lambda self: self._get_raw_labels().dtype == np.int64)

    @property
    def resolution(self):
        assert len(self.image_shape) == 3 and self.image_shape[1] == self.image_shape[2]
        return self.image_shape[1]

    @property
    def label_shape(self):
        if self._label_shape is None:
            labels = self._get_raw_labels()
            self._label_shape = [int(np.max(labels)) + 1] if labels.dtype == np.int64 else list(labels.shape[1:])
        return list(self._label_shape)

    @property
    def label_dim(self):
        assert len(self.label_shape) == 1
        return self.label_shape[0]


@datasets.add_to_registry("image_folder")
class ImageFolderDataset(Dataset):
    def __init__(self, path='', resolution=None, use_labels=False, max_size=None, xflip=False, random_seed=0):
        self._path, self._zipfile = path, None
        if os.path.isdir(path):
            self._type = 'dir'
            self._all_fnames = {os.path.relpath(os.path.join(root, f), start=path) for root, _d, files in os.walk(path) for f in files}
        elif os.path.splitext(path)[1].lower() == '.zip':
            self._type = 'zip'
            self._all_fnames = set(self._zip().namelist())
        else:
            raise IOError('Path must point to a directory or zip')
        PIL.Image.init()
        self._image_fnames = sorted(f for f in self._all_fnames if os.path.splitext(f)[1].lower() in PIL.Image.EXTENSION)
        if not self._image_fnames:
            raise IOError('No image files found in the specified path')
        raw_shape = [len(self._image_fnames)] + list(self._load_raw_image(0).shape)
        if resolution is not None and (raw_shape[2] != resolution or raw_shape[3] != resolution):
            raise IOError('Image files do not match the specified resolution')
        super().__init__(name=os.path.splitext(os.path.basename(path.rstrip('/')))[0], raw_shape=raw_shape, max_size=max_size,
                         use_labels=use_labels, xflip=xflip, random_seed=random_seed)

    def _zip(self):
        if self._zipfile is None:
            self._zipfile = zipfile.ZipFile(self._path)
        return self._zipfile

    def _open(self, fname):
        return open(os.path.join(self._path, fname), 'rb') if self._type == 'dir' else self._zip().open(fname, 'r')

    def close(self):
        try:
            if self._zipfile is not None:
                self._zipfile.close()
        finally:
            self._zipfile = None

    def __getstate__(self):
        return dict(super().__getstate__(), _zipfile=None)

    def _load_raw_image(self, raw_idx):
        with self._open(self._image_fnames[raw_idx]) as f:
            image = np.array(PIL.Image.open(f))
        if image.ndim == 2:
            image = image[:, :, np.newaxis]
        return image.transpose(2, 0, 1)                     # HWC -> CHW

    def _load_raw_labels(self):
        if 'dataset.json' not in self._all_fnames:
            return None
        with self._open('dataset.json') as f:
            labels = json.load(f)['labels']
        if labels is None:
            return None
        labels = dict(labels)
        labels = np.array([labels[f.replace('\\\\', '/')] for f in self._image_fnames])
        return labels.astype({1: np.int64, 2: np.float32}[labels.ndim])
