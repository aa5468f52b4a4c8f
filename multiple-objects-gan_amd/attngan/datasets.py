"""Data side of the coco-attngan train step (mirror of code/coco/attngan/datasets.py).

`prepare_data` keeps the reference's output contract (datasets.py:28-68): sort the minibatch by caption
length (descending), move it to the device, return
    [real_imgs[3], captions (B,T), sorted_cap_lens (B,), class_ids (numpy), keys, [tm, tmi], label_one_hot].
`crop_imgs` restates the crop/flip + bbox rescale/clamp rules (datasets.py:95-137).
`TextDataset` reads the reference's pickles (captions.pickle, <split>/filenames.pickle, bboxes.pickle,
labels.pickle) and JPEGs with PIL + numpy only (torchvision is not a dependency here);
`SyntheticTextDataset` yields samples of the same structure without any file (benchmarks, smoke runs).
"""
import os
import pickle

import numpy as np
import torch
import torch.utils.data as data

from . import synthetic
from .miscc.config import cfg
from .miscc.utils import compute_transformation_matrix, compute_transformation_matrix_inverse


def prepare_data(batch, device=None, eval=False):
    if eval:
        imgs, captions, captions_lens, class_ids, keys, transformation_matrices, label, bbox = batch
    else:
        imgs, captions, captions_lens, class_ids, keys, transformation_matrices, label = batch
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if cfg.CUDA else torch.device("cpu")
    sorted_cap_lens, order = torch.sort(captions_lens, 0, True)
    real_imgs = [im[order].to(device, non_blocking=True) for im in imgs]
    captions = captions[order].squeeze(-1).to(device)
    class_ids = class_ids[order].numpy()
    tm = transformation_matrices[0][order].to(device)
    tmi = transformation_matrices[1][order].to(device)
    label = label[order].to(device)
    keys = [keys[i] for i in order.numpy()]
    out = [real_imgs, captions, sorted_cap_lens.to(device), class_ids, keys, [tm, tmi], label]
    if eval:
        out.append(bbox[order])
    return out


def prepare_data_raw(batch, feeder, rng=np.random, eval=False):
    """prepare_data for TextDataset(raw=True) batches: per sample the crop offsets / flip are drawn and the boxes rescaled
    on the host (feeder.draw_crop = the arithmetic of crop_imgs), the images are cropped / flipped / resampled /
    normalised on the device (feeder.DeviceFeeder).  Same return structure as prepare_data."""
    from .feeder import draw_crop
    u8, captions, captions_lens, class_ids, keys, bboxes, label = batch
    B = u8.shape[0]
    params, scaled = [], []
    for b in range(B):
        p, sb = draw_crop(bboxes[b].numpy(), rng)
        params.append(p)
        scaled.append(sb)
    slot = feeder.upload(u8, np.asarray(params, dtype=np.int32))        # starts the H2D copy; the rest overlaps it
    scaled = torch.from_numpy(np.stack(scaled).astype(np.float32))
    tmi = compute_transformation_matrix_inverse(scaled.view(-1, 4)).view(B, -1, 2, 3)
    tm = compute_transformation_matrix(scaled.view(-1, 4)).view(B, -1, 2, 3)
    device = feeder.device
    sorted_cap_lens, order = torch.sort(captions_lens, 0, True)
    imgs = feeder.process(slot)
    od = order.to(device)
    real_imgs = [im.index_select(0, od) for im in imgs]
    keys = [keys[i] for i in order.numpy()]
    out = [real_imgs, captions[order].squeeze(-1).to(device), sorted_cap_lens.to(device), class_ids[order].numpy(), keys,
           [tm[order].to(device), tmi[order].to(device)], label[order].to(device)]
    if eval:
        out.append(scaled[order])
    return out


def crop_imgs(image, bbox, max_objects=3, rng=np.random):
    """image (3,268,268) float tensor, bbox (max_objects,4) relative (x,y,w,h) or -1 -> random 256 crop,
    random horizontal flip, bbox rescaled to the crop with the reference's clamp (x+w > 0.999 -> w = 1-x-0.001)."""
    ori_size, imsize = 268, 256
    flip = rng.random() < 0.5
    margin = ori_size - imsize
    h1 = int(np.floor(margin * rng.random()))
    w1 = int(np.floor(margin * rng.random()))
    out = np.full_like(bbox, -1.0)
    for k in range(max_objects):
        b = bbox[k]
        if b[0] == -1:
            break
        x = max(b[0] * float(ori_size) - h1, 0) / float(imsize)
        y = max(b[1] * float(ori_size) - w1, 0) / float(imsize)
        w = min(float(ori_size) / imsize * b[2], 1.0)
        if x + w > 0.999:
            w = 1.0 - x - 0.001
        h = min(float(ori_size) / imsize * b[3], 1.0)
        if y + h > 0.999:
            h = 1.0 - y - 0.001
        if flip:
            x = 1.0 - x - w
        out[k] = [x, y, w, h]
    img = image[:, w1:w1 + imsize, h1:h1 + imsize]
    if flip:
        img = torch.flip(img, dims=[2])
    return img, out


def _to_tensor(pil_img):
    a = np.asarray(pil_img, dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)


def _multi_scale(img256, imsize):
    """(3,256,256) in [0,1] -> list of normalised ([-1,1]) images at the branch resolutions."""
    from PIL import Image
    ret = []
    pil = Image.fromarray((img256.permute(1, 2, 0).numpy() * 255.0 + 0.5).astype(np.uint8))
    for i, s in enumerate(imsize):
        im = pil if i == len(imsize) - 1 else pil.resize((s, s), Image.BILINEAR)
        ret.append((_to_tensor(im) - 0.5) / 0.5)
    return ret


class _Base(data.Dataset):
    max_objects = 3

    def _matrices(self, bbox):
        b = torch.from_numpy(np.asarray(bbox, dtype=np.float32)).view(-1, 4)
        tmi = compute_transformation_matrix_inverse(b).view(self.max_objects, 2, 3)
        tm = compute_transformation_matrix(b).view(self.max_objects, 2, 3)
        return tm, tmi

    @staticmethod
    def _one_hot(label):
        return synthetic.one_hot_labels(np.asarray(label).reshape(-1))


class TextDataset(_Base):
    def __init__(self, data_dir, img_dir, split='train', base_size=64, transform=None, target_transform=None,
                 eval=False, raw=False):
        """raw=True: the workers stop after the JPEG decode + resize; a sample then carries the 268x268 u8 image and the
        UNSCALED boxes, and crop / flip / multi-scale / normalise run on the device (feeder.DeviceFeeder via
        prepare_data_raw)."""
        self.embeddings_num = cfg.TEXT.CAPTIONS_PER_IMAGE
        self.img_dir, self.data_dir, self.eval, self.raw = img_dir, data_dir, eval, raw
        self.split_dir = os.path.join(data_dir, split)
        self.imsize = [base_size << i for i in range(cfg.TREE.BRANCH_NUM)]
        with open(os.path.join(self.split_dir, 'bboxes.pickle'), 'rb') as f:
            self.bbox = np.array(pickle.load(f))
        with open(os.path.join(self.split_dir, 'labels.pickle'), 'rb') as f:
            self.labels = np.array(pickle.load(f))
        with open(os.path.join(data_dir, 'captions.pickle'), 'rb') as f:
            x = pickle.load(f)
        self.ixtoword, self.wordtoix = x[2], x[3]
        self.n_words = len(self.ixtoword)
        self.captions = x[0] if split == 'train' else x[1]
        with open('%s/%s/filenames.pickle' % (data_dir, split), 'rb') as f:
            self.filenames = pickle.load(f)
        cls_path = self.split_dir + '/class_info.pickle'
        if os.path.isfile(cls_path):
            with open(cls_path, 'rb') as f:
                self.class_id = pickle.load(f)
        else:
            self.class_id = np.arange(len(self.filenames))

    def get_caption(self, sent_ix):
        cap = np.asarray(self.captions[sent_ix]).astype('int64')
        T = cfg.TEXT.WORDS_NUM
        x = np.zeros((T, 1), dtype='int64')
        n = len(cap)
        if n <= T:
            x[:n, 0] = cap
        else:
            ix = np.sort(np.random.permutation(n)[:T])
            x[:, 0] = cap[ix]
            n = T
        return x, n

    def __getitem__(self, index):
        from PIL import Image
        key = self.filenames[index]
        img = Image.open('%s/%s.jpg' % (self.img_dir, key)).convert('RGB').resize((268, 268), Image.BILINEAR)
        if self.raw:
            sent_ix = np.random.randint(0, self.embeddings_num)
            caps, cap_len = self.get_caption(index * self.embeddings_num + sent_ix)
            return (torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()), caps, cap_len, self.class_id[index], key,
                    torch.from_numpy(np.asarray(self.bbox[index], dtype=np.float32).copy()),
                    self._one_hot(self.labels[index]))
        img, bbox_scaled = crop_imgs(_to_tensor(img), self.bbox[index])
        imgs = _multi_scale(img, self.imsize)
        tms = self._matrices(bbox_scaled)
        label = self._one_hot(self.labels[index])
        sent_ix = np.random.randint(0, self.embeddings_num)
        caps, cap_len = self.get_caption(index * self.embeddings_num + sent_ix)
        if self.eval:
            return imgs, caps, cap_len, self.class_id[index], key, tms, label, bbox_scaled
        return imgs, caps, cap_len, self.class_id[index], key, tms, label

    def __len__(self):
        return len(self.filenames)


class SyntheticTextDataset(_Base):
    """Same sample structure as TextDataset, generated: images U(-1,1), captions uniform in
    [1, n_words), 2-3 boxes per image with the reference clamp rules, labels uniform in [0,80)."""

    def __init__(self, length=1024, n_words=synthetic.VOCAB, seed=0, eval=False):
        self.length, self.n_words, self.seed, self.eval = length, n_words, seed, eval
        self.ixtoword = {i: ('<end>' if i == 0 else 'w%d' % i) for i in range(n_words)}
        self.imsize = [cfg.TREE.BASE_SIZE << i for i in range(cfg.TREE.BRANCH_NUM)]

    def __getitem__(self, index):
        rng = np.random.RandomState(self.seed * 1000003 + index)
        T = cfg.TEXT.WORDS_NUM
        imgs = [torch.from_numpy(rng.uniform(-1, 1, (3, s, s)).astype(np.float32)) for s in self.imsize]
        n = int(rng.randint(min(5, max(1, T // 2)), T + 1))
        caps = np.zeros((T, 1), dtype='int64')
        caps[:n, 0] = rng.randint(1, self.n_words, n)
        bbox, labels = synthetic.make_bboxes(rng, 1)
        tms = self._matrices(bbox[0])
        if self.eval:                                    # datasets.py:374-375: the scaled boxes ride along for sample()
            return imgs, caps, n, index, 'synthetic_%06d' % index, tms, self._one_hot(labels[0]), bbox[0]
        return imgs, caps, n, index, 'synthetic_%06d' % index, tms, self._one_hot(labels[0])

    def __len__(self):
        return self.length
