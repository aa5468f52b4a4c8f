"""`GANTrainer.sample` of the three StackGAN-family trees (code/coco/stackgan/trainer.py:287-419, code/clevr/trainer.py:198-295,
code/multi-mnist/trainer.py:208-343; VERDICT r4 "missing" item 5): load the generator checkpoint (cfg.NET_G), put it in
eval mode, and for `num_samples` test items write one PNG with the real image, nine samples for nine noise vectors, the
boxes drawn in, and (clevr / mnist) a second row with the label text -- into `<NET_G without .pth>_<suffix>/`.

Same inputs per image as the reference: the conditioning of ONE test item repeated nine times, noise ~ N(0, 1) drawn per image,
`np.random.randint` for the item index (coco, mnist) or the data loader's order (clevr).  The generator runs through the same HIP
modules as in training (BatchNorm on its running statistics); everything else here is host-side PIL / numpy."""
import os

import numpy as np
import torch

from ..attngan.miscc.utils import mkdir_p
from ..attngan.synthetic import bbox_to_theta, one_hot_labels
from . import t7
from .datasets import _load_pickle, load_validation_data
from .logging_utils import save_image
from .synthetic import _theta64

CLEVR_SHAPE_NAMES = {0: "cube", 1: "cylinder", 2: "sphere", 3: "empty"}
CLEVR_COLOR_NAMES = {0: "gray", 1: "red", 2: "blue", 3: "green", 4: "brown", 5: "purple", 6: "cyan", 7: "yellow", 8: "empty"}


def _to_tensor(img):
    a = np.array(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)


def _draw_boxes(data_img, boxes, imsize, stop_on_y=False, shrink=False):
    """the four box edges set to 1 in the first ten images (S/trainer.py:401-411; M/trainer.py:311-326 shrinks boxes that
    reach the border, C/trainer.py:254-257 also stops at y == -1)."""
    for b in boxes:
        x, y, w, h = tuple(int(imsize * float(v)) for v in b)
        w = imsize - 1 if w > imsize - 1 else w
        h = imsize - 1 if h > imsize - 1 else h
        if shrink:
            while x + w >= imsize:
                x -= 1
                w -= 1
            while y + h >= imsize:
                y -= 1
                h -= 1
        if x <= -1 or (stop_on_y and y <= -1):
            break
        x2, y2 = min(x + w, imsize - 1), min(y + h, imsize - 1)    # (a box that ends ON the border indexes past it in the reference)
        data_img[:10, :, y, x:x + w] = 1
        data_img[:10, :, y:y + h, x] = 1
        data_img[:10, :, y2, x:x + w] = 1
        data_img[:10, :, y:y + h, x2] = 1


def _text_row(text, imsize):
    """ten (1, imsize, imsize) tiles of a white strip with `text` written at (10, 10) (C/trainer.py:279-290)."""
    from PIL import Image, ImageDraw
    strip = Image.new('L', (imsize * 10, imsize), color='white')
    ImageDraw.Draw(strip).text((10, 10), text)
    tiles = torch.chunk(_to_tensor(strip), 10, 2)
    return torch.cat([t.reshape(1, 1, imsize, imsize) for t in tiles], 0)


def _save_dir(cfg, suffix):
    save_dir = cfg.NET_G[:cfg.NET_G.find('.pth')] + suffix
    print("saving to:", save_dir)
    mkdir_p(save_dir)
    return save_dir


def _safe_name(caption):
    return "".join(c if c not in '/\\\0' else "_" for c in caption)[:200]


@torch.no_grad()
def sample_coco(trainer, datapath, num_samples=25, stage=1, draw_bbox=True, max_objects=3):
    from PIL import Image
    cfg, dev = trainer.cfg, trainer.device
    nets = trainer.load_network_stageI() if stage == 1 else trainer.load_network_stageII()
    if nets is None:
        return None
    netG = nets[0].eval()
    t_file = t7.load(datapath + "val_captions.t7")
    captions_list = t_file["raw_txt"]
    embeddings = np.concatenate([np.asarray(e).reshape(-1, np.asarray(e).shape[-1]) for e in t_file["fea_txt"]], axis=0)
    num_embeddings = len(captions_list)
    label, bbox = load_validation_data(datapath)
    filenames = _load_pickle(os.path.join(datapath, 'filenames.pickle'))
    print('Successfully load sentences from: ', datapath)
    print('Total number of sentences:', num_embeddings)
    save_dir = _save_dir(cfg, "_visualize_bbox")
    K = max_objects
    bbox_ = bbox.clone()
    # the reference feeds the same (stage-I scaled) boxes of the test split to both stages (S/trainer.py:316-338)
    tm, tmi = bbox_to_theta(bbox.reshape(-1, 4))
    tm, tmi = tm.view(num_embeddings, K, 2, 3).to(dev), tmi.view(num_embeddings, K, 2, 3).to(dev)
    label_one_hot = one_hot_labels(label.reshape(num_embeddings, K)).to(dev)
    imsize = 64 if stage == 1 else 256
    written = []
    for count in range(num_samples):
        index = int(np.random.randint(0, num_embeddings, 1)[0])
        img = Image.open(cfg.IMG_DIR + "/" + filenames[index] + ".jpg").convert('RGB').resize((imsize, imsize), Image.LANCZOS)
        val_image = (_to_tensor(img).view(1, 3, imsize, imsize) - 0.5) * 2
        txt = torch.from_numpy(np.reshape(embeddings[index], (1, -1)).repeat(9, 0)).float().to(dev)
        tmi_b = tmi[index].view(1, K, 2, 3).repeat(9, 1, 1, 1)
        lab_b = label_one_hot[index].view(1, K, 81).repeat(9, 1, 1)
        noise = torch.randn(9, cfg.Z_DIM, device=dev)
        if stage == 1:
            _, fake_imgs, _, _, _ = netG(txt, noise, tmi_b, lab_b)
        else:
            tm_b = tm[index].view(1, K, 2, 3).repeat(9, 1, 1, 1)
            _, fake_imgs, _, _, _ = netG(txt, noise, tmi_b, tm_b, tmi_b, lab_b)
        data_img = torch.zeros(10, 3, imsize, imsize)
        data_img[0] = val_image
        data_img[1:10] = fake_imgs.float().cpu()
        if draw_bbox:
            _draw_boxes(data_img, bbox_[index, :3], imsize)
        written.append(save_image(data_img, '{}/{}.png'.format(save_dir, _safe_name(captions_list[index])), nrow=10))
    print("Saved {} files to {}".format(len(written), save_dir))
    return written


@torch.no_grad()
def sample_clevr(trainer, data_loader, num_samples=25, draw_bbox=True, max_objects=4):
    cfg, dev = trainer.cfg, trainer.device
    netG = trainer.load_network_stageI()[0].eval()
    save_dir = _save_dir(cfg, "_samples_" + str(max_objects) + "_objects")
    imsize, K, written = 64, max_objects, []
    for data in data_loader:
        if len(written) == num_samples:
            break
        real_img, (_, tmi), label_one_hot, bbox = data
        tmi_b = tmi.float().to(dev).view(1, K, 2, 3).repeat(9, 1, 1, 1)
        lab_b = label_one_hot.float().to(dev).view(1, K, 13).repeat(9, 1, 1)
        noise = torch.randn(9, cfg.Z_DIM, device=dev)
        fake_imgs = netG(noise, tmi_b, lab_b)
        data_img = torch.zeros(20, 3, imsize, imsize)
        data_img[0] = real_img[0]
        data_img[1:10] = fake_imgs.float().cpu()
        if draw_bbox:
            _draw_boxes(data_img, bbox[0, :K], imsize, stop_on_y=True)
        lab = lab_b[0].cpu().numpy()
        shape, color = np.argmax(lab[:, :4], axis=1), np.argmax(lab[:, 4:], axis=1)
        text = ", ".join(CLEVR_COLOR_NAMES[int(color[i])] + " " + CLEVR_SHAPE_NAMES[int(shape[i])] for i in range(K))
        data_img[10:] = _text_row(text, imsize)
        written.append(save_image(data_img, '{}/vis_{}.png'.format(save_dir, len(written)), nrow=10))
    print("Saved {} files to {}".format(len(written), save_dir))
    return written


@torch.no_grad()
def sample_mnist(trainer, datapath, num_samples=25, draw_bbox=True, num_digits_per_img=3, change_bbox_size=False):
    from PIL import Image
    cfg, dev = trainer.cfg, trainer.device
    img_dir = os.path.join(datapath, "normal", "imgs/")
    netG = trainer.load_network_stageI()[0].eval()
    label, bbox = load_validation_data(datapath, tree="mnist")
    test_set_size = bbox.shape[0]                    # (the reference hard-codes its test split's 10000)
    K = num_digits_per_img
    if K < 3:
        label, bbox = label[:, :K, :], bbox[:, :K, ...]
    elif K > 3:                                      # extra digits with random identities and boxes (M/trainer.py:222-244)
        extra = np.eye(10)[np.random.randint(0, 10, size=(bbox.shape[0], K - 3)).reshape(-1)].reshape(bbox.shape[0], K - 3, 10)
        labels_new = np.zeros((label.shape[0], K, 10))
        labels_new[:, :3, :], labels_new[:, 3:, :] = label, extra
        label = torch.from_numpy(labels_new)
        bx, by = np.random.random((bbox.shape[0], K - 3, 1)), np.random.random((bbox.shape[0], K - 3, 1))
        bw = np.random.randint(10, 20, size=(bbox.shape[0], K - 3, 1)) / 64.0
        bh = np.random.randint(16, 20, size=(bbox.shape[0], K - 3, 1)) / 64.0
        bbox_new = np.zeros([bbox.shape[0], K, 4])
        bbox_new[:, :3, :], bbox_new[:, 3:, :] = bbox, np.concatenate((bx, by, bw, bh), axis=2)
        bbox = torch.from_numpy(bbox_new)
    if change_bbox_size:
        bbox_idx = np.random.randint(0, bbox.shape[1])
        scale_x, scale_y = np.random.random(bbox.shape[0]), np.random.random(bbox.shape[0])
        scale_x[scale_x < 0.5] = 0.5
        scale_y[scale_y < 0.5] = 0.5
        bbox[:, bbox_idx, 2] *= torch.from_numpy(scale_x)
        bbox[:, bbox_idx, 3] *= torch.from_numpy(scale_y)
    filenames = _load_pickle(os.path.join(datapath, "normal", 'filenames.pickle'))
    suffix = "_samples_" + str(K) + "_digits" + ("_change_bbox_size" if change_bbox_size else "")
    save_dir = _save_dir(cfg, suffix)
    bbox_ = bbox.clone()
    _, tmi = _theta64(bbox.reshape(-1, 4))
    tmi = tmi.view(test_set_size, K, 2, 3).to(dev)
    label_one_hot = label.float().to(dev)
    imsize, written = 64, []
    for count in range(num_samples):
        index = int(np.random.randint(0, test_set_size, 1)[0])
        img = Image.open(img_dir + filenames[index].split("/")[-1])
        val_image = (_to_tensor(img).view(1, 1, imsize, imsize) - 0.5) * 2
        tmi_b = tmi[index].view(1, K, 2, 3).repeat(9, 1, 1, 1)
        lab_b = label_one_hot[index].view(1, K, 10).repeat(9, 1, 1)
        noise = torch.randn(9, cfg.Z_DIM, device=dev)
        _, fake_imgs = netG(noise, tmi_b, lab_b, K)
        data_img = torch.zeros(20, 1, imsize, imsize)
        data_img[0] = val_image
        data_img[1:10] = fake_imgs.float().cpu()
        if draw_bbox:
            _draw_boxes(data_img, bbox_[index, :K], imsize, shrink=True)
        digits = np.argmax(lab_b[0].cpu().numpy(), axis=1)
        data_img[10:] = _text_row(", ".join(str(int(d)) for d in digits), imsize)
        written.append(save_image(data_img, '{}/vis_{}.png'.format(save_dir, count), nrow=10))
    print("Saved {} files to {}".format(len(written), save_dir))
    return written
