"""Real-data `TextDataset`s of the three StackGAN-family trees (VERDICT r4 "missing" item 5): the
file formats, the random crop / flip with its box rescale, and the item tuples of

  coco   code/coco/stackgan/miscc/datasets.py:25-217   <split>/{filenames,bboxes,labels,char-CNN-RNN-embeddings}.pickle + JPEGs;
         item (img, bbox | [bbox_stage1, bbox_stage2], label, embedding)
  clevr  code/clevr/miscc/datasets.py:44-145           <split>/scenes/*.json + <split>/images/*;
         item (img, (theta, theta_inv), label_one_hot(4, 4 + 9), bbox)
  mnist  code/multi-mnist/miscc/datasets.py:25-86      <split>/normal/{filenames,bboxes,labels}.pickle + imgs/;
         item (img, bbox float64, label)

The default-collated minibatch of each is what `trainer_base.GANTrainerBase.prepare_batch` (the reference trainers' loop
prologue) unpacks.  torchvision is not installed: `image_transform` restates Resize (PIL bilinear) + ToTensor + Normalize(0.5, 0.5)
of the reference's main.py files (S/main.py:84-87, C/main.py:82-84, M/main.py:79-81).  The random draws keep the reference's
generators and order -- `random.random()` for the flip, then two `np.random.random()` for the crop offsets -- so a seeded run
reproduces the reference's crops (tests/golden/stackgan_data.npz is generated from the reference's own crop_imgs)."""
import glob
import json
import os
import pickle
import random

import numpy as np
import torch
import torch.utils.data as data

from ..attngan.synthetic import bbox_to_theta


def image_transform(resize=None):
    """transforms.Compose([Resize((r, r))?, ToTensor(), Normalize(.5, .5)]): PIL image -> float32 (C, H, W) in [-1, 1]."""
    from PIL import Image

    def apply(img):
        if resize is not None:
            img = img.resize((resize, resize), Image.BILINEAR)
        a = np.array(img, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)
        return t.sub_(0.5).div_(0.5)
    return apply


def _load_pickle(path):
    with open(path, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:                 # pickles written by python 2 (the reference's era)
            f.seek(0)
            return pickle.load(f, encoding="latin1")


def _rescale_box(b, ori, size, h1, w1, flip):
    """one box through the crop at column offset h1 / row offset w1 (S/miscc/datasets.py:116-131)."""
    x_new = max(b[0] * float(ori) - h1, 0) / float(size)
    y_new = max(b[1] * float(ori) - w1, 0) / float(size)
    width_new = min((float(ori) / size) * b[2], 1.0)
    if x_new + width_new > 0.999:
        width_new = 1.0 - x_new - 0.001
    height_new = min((float(ori) / size) * b[3], 1.0)
    if y_new + height_new > 0.999:
        height_new = 1.0 - y_new - 0.001
    if flip:
        x_new = 1.0 - x_new - width_new
    return [x_new, y_new, width_new, height_new]


def crop_imgs(image, bbox, imsize, stage=1, max_objects=3):
    """S/miscc/datasets.py:100-183: random imsize crop of the (3, ori, ori) image, random horizontal flip, the boxes rescaled into
    the crop (absent objects -- the first x == -1 ends the list -- stay -1).  Stage 2 returns TWO box sets from the same offsets:
    76 -> 64 for the frozen stage-I generator, ori -> imsize for stage II."""
    ori_size = image.shape[1]
    flip_img = random.random() < 0.5
    img_crop = ori_size - imsize
    h1 = int(np.floor(img_crop * np.random.random()))
    w1 = int(np.floor(img_crop * np.random.random()))
    sets = [np.zeros_like(bbox) for _ in range(2 if stage != 1 else 1)]
    for s in sets:
        s[...] = -1.0
    for idx in range(max_objects):
        b = bbox[idx]
        if b[0] == -1:
            break
        if stage == 1:
            sets[0][idx] = _rescale_box(b, ori_size, imsize, h1, w1, flip_img)
        else:
            sets[0][idx] = _rescale_box(b, 76, 64, h1, w1, flip_img)
            sets[1][idx] = _rescale_box(b, ori_size, imsize, h1, w1, flip_img)
    cropped = image[:, w1: w1 + imsize, h1: h1 + imsize]
    if flip_img:
        cropped = torch.flip(cropped, dims=[2])
    return cropped, (sets[0] if stage == 1 else sets)


class CocoTextDataset(data.Dataset):
    """code/coco/stackgan/miscc/datasets.py TextDataset (same constructor arguments)."""

    EMBEDDINGS = {"cnn-rnn": "char-CNN-RNN-embeddings.pickle", "cnn-gru": "char-CNN-GRU-embeddings.pickle",
                  "skip-thought": "skip-thought-embeddings.pickle"}

    def __init__(self, data_dir, img_dir, imsize, split='train', embedding_type='cnn-rnn', transform=None, crop=True, stage=1):
        self.transform, self.imsize, self.crop, self.stage = transform, imsize, crop, stage
        self.data_dir, self.img_dir = data_dir, img_dir
        self.split_dir = os.path.join(data_dir, split)
        self.max_objects = 3
        self.filenames = _load_pickle(os.path.join(self.split_dir, 'filenames.pickle'))
        print('Load filenames from: %s (%d)' % (os.path.join(self.split_dir, 'filenames.pickle'), len(self.filenames)))
        self.bboxes = np.array(_load_pickle(os.path.join(self.split_dir, 'bboxes.pickle')))
        self.labels = np.array(_load_pickle(os.path.join(self.split_dir, 'labels.pickle')))
        self.embeddings = np.array(_load_pickle(os.path.join(self.split_dir, self.EMBEDDINGS[embedding_type])))

    def get_img(self, img_path):
        from PIL import Image
        img = Image.open(img_path).convert('RGB')
        return self.transform(img) if self.transform is not None else img

    def __getitem__(self, index):
        key = self.filenames[index]
        img = self.get_img(self.img_dir + "/" + key + ".jpg")
        bbox, label = self.bboxes[index], self.labels[index]
        embeddings = self.embeddings[index, :, :]
        embedding = embeddings[random.randint(0, embeddings.shape[0] - 1), :]
        if self.crop:
            img, bbox = crop_imgs(img, bbox, self.imsize, self.stage, self.max_objects)
        return img, bbox, label, embedding

    def __len__(self):
        return len(self.filenames)


CLEVR_SHAPES = {"cube": 0, "cylinder": 1, "sphere": 2}
CLEVR_COLORS = {"gray": 0, "red": 1, "blue": 2, "green": 3, "brown": 4, "purple": 5, "cyan": 6, "yellow": 7}


class ClevrTextDataset(data.Dataset):
    """code/clevr/miscc/datasets.py TextDataset: one scene json per item; the flip is drawn per image load."""

    def __init__(self, data_dir, imsize, split='train', transform=None):
        self.transform, self.imsize, self.data_dir = transform, imsize, data_dir
        self.split_dir = os.path.join(data_dir, split)
        self.img_dir = os.path.join(self.split_dir, "images")
        self.scene_dir = os.path.join(self.split_dir, "scenes")
        self.max_objects = 4
        self.filenames = [f for f in glob.glob(self.scene_dir + '/*.json')]
        print('Load scenes from: %s (%d)' % (self.scene_dir, len(self.filenames)))

    def get_img(self, img_path):
        from PIL import Image
        img = Image.open(img_path).convert('RGB')
        if self.transform is not None:
            img = self.transform(img)
        flip_img = random.random() < 0.5
        if flip_img:
            img = torch.flip(img, dims=[2])
        return img, flip_img

    @staticmethod
    def label_one_hot(label, dim):
        """-1 (no object) -> the last class; (K, 1) indices -> (K, dim) one-hot (C/miscc/datasets.py:104-112)."""
        labels = torch.from_numpy(label).long()
        labels[labels < 0] = dim - 1
        return torch.zeros(labels.shape[0], dim).scatter_(1, labels, 1).float()

    def calc_transformation_matrix(self, bbox):
        tm, tmi = bbox_to_theta(torch.from_numpy(bbox).view(-1, 4))
        return tm.view(self.max_objects, 2, 3), tmi.view(self.max_objects, 2, 3)

    def __getitem__(self, index):
        with open(self.filenames[index], "rb") as f:
            scene = json.load(f)
        img, flip_img = self.get_img(self.img_dir + "/" + scene["image_filename"])
        K = self.max_objects
        bbox = np.full((K, 4), -1.0, dtype=np.float32)
        label_shape, label_color = np.full(K, -1.0), np.full(K, -1.0)
        for idx, obj in enumerate(scene["objects"]):
            bbox[idx, :] = obj["bbox"]
            label_shape[idx] = CLEVR_SHAPES[obj["shape"]]
            label_color[idx] = CLEVR_COLORS[obj["color"]]
        bbox = bbox / float(self.imsize)
        label = torch.cat((self.label_one_hot(np.expand_dims(label_shape, 1), 4),
                           self.label_one_hot(np.expand_dims(label_color, 1), 9)), 1)
        if flip_img:
            bbox[:, 0] = 1.0 - bbox[:, 0] - bbox[:, 2]
        return img, self.calc_transformation_matrix(bbox), label, bbox

    def __len__(self):
        return len(self.filenames)


class MnistTextDataset(data.Dataset):
    """code/multi-mnist/miscc/datasets.py TextDataset: float64 boxes, one-hot labels as pickled."""

    def __init__(self, data_dir, imsize, split='train', transform=None, crop=False):
        self.transform, self.imsize, self.crop, self.data_dir = transform, imsize, crop, data_dir
        self.split_dir = os.path.join(data_dir, split, "normal")
        self.img_dir = self.split_dir + "/imgs/"
        self.max_objects = 3
        self.filenames = _load_pickle(os.path.join(self.split_dir, 'filenames.pickle'))
        print('Load filenames from: %s (%d)' % (os.path.join(self.split_dir, 'filenames.pickle'), len(self.filenames)))
        self.bboxes = np.array(_load_pickle(os.path.join(self.split_dir, 'bboxes.pickle')), dtype=np.double)
        self.labels = np.array(_load_pickle(os.path.join(self.split_dir, 'labels.pickle')))

    def get_img(self, img_path):
        from PIL import Image
        img = Image.open(img_path)
        return self.transform(img) if self.transform is not None else img

    def __getitem__(self, index):
        key = self.filenames[index].split("/")[-1]
        return self.get_img(self.split_dir + "/imgs/" + key), self.bboxes[index].astype(np.double), self.labels[index]

    def __len__(self):
        return len(self.filenames)


def load_validation_data(datapath, tree="coco"):
    """labels, boxes of a test split as tensors (S/miscc/utils.py:54-64: <datapath>{bboxes,labels}.pickle;
    M/miscc/utils.py:59-68: <datapath>/normal/...)."""
    if tree == "mnist":
        bb, lb = os.path.join(datapath, "normal", "bboxes.pickle"), os.path.join(datapath, "normal", "labels.pickle")
    else:
        bb, lb = datapath + "bboxes.pickle", datapath + "labels.pickle"
    return torch.from_numpy(np.array(_load_pickle(lb))), torch.from_numpy(np.array(_load_pickle(bb)))
