"""Attention-map grids of the AttnGAN trainer (host side, PIL/numpy):
`build_super_images` (miscc/utils.py:88-211) is what `save_img_results` (trainer.py:206-248) writes every few thousand
iterations -- per sample a caption strip, the row of up-scaled word-attention maps and the row of the image blended with
each map; `build_super_images2` (utils.py:214-316) is the top-K variant the sampling code uses.

Two dependencies of the reference are absent here and restated:
  * `skimage.transform.pyramid_expand` (scikit-image 0.14): bilinear up-scaling by `upscale` followed by a Gaussian
    smoothing of the spatial axes (`scipy.ndimage.gaussian_filter`, mode 'reflect'; the channel axis is not smoothed).
    scikit-image is not vendored in the reference and not installed in this image, so this restatement is NOT pinned
    against the original (tests pin everything around it by handing the same function to the reference's code);
  * the caption font 'Pillow/Tests/fonts/FreeMono.ttf': used when PIL can load it, PIL's built-in font otherwise.
"""
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image, ImageDraw, ImageFont
from scipy import ndimage

from .config import cfg

COLOR_DIC = {0: [128, 64, 128], 1: [244, 35, 232], 2: [70, 70, 70], 3: [102, 102, 156], 4: [190, 153, 153],
             5: [153, 153, 153], 6: [250, 170, 30], 7: [220, 220, 0], 8: [107, 142, 35], 9: [152, 251, 152],
             10: [70, 130, 180], 11: [220, 20, 60], 12: [255, 0, 0], 13: [0, 0, 142], 14: [119, 11, 32],
             15: [0, 60, 100], 16: [0, 80, 100], 17: [0, 0, 230], 18: [0, 0, 70], 19: [0, 0, 0]}
FONT_MAX = 50          # height of the caption strip of one sample (pixels) = the font size


def pyramid_expand(image, upscale=2, sigma=None):
    """(H, W, C) float array -> (H*upscale, W*upscale, C): bilinear resize, then Gaussian smoothing (sigma defaults to
    2*upscale/6 like scikit-image; the reference passes sigma=20)."""
    image = np.asarray(image, dtype=np.float64)
    if sigma is None:
        sigma = 2 * upscale / 6.0
    t = torch.from_numpy(image).permute(2, 0, 1)[None]
    t = F.interpolate(t, scale_factor=upscale, mode="bilinear", align_corners=False)
    out = t[0].permute(1, 2, 0).numpy()
    return ndimage.gaussian_filter(out, sigma=(sigma, sigma, 0), mode="reflect")


def _font():
    for path in ('Pillow/Tests/fonts/FreeMono.ttf', 'FreeMono.ttf', 'DejaVuSansMono.ttf'):
        try:
            return ImageFont.truetype(path, FONT_MAX)
        except (OSError, IOError):
            continue
    return ImageFont.load_default()


def drawCaption(convas, captions, ixtoword, vis_size, off1=2, off2=2):
    """Writes '<j>:<first 6 letters>' of word j of caption i at column (j+off1)*(vis_size+off2), row i*FONT_MAX.
    -> (PIL image, list of word lists)."""
    img_txt = Image.fromarray(convas)
    fnt = _font()
    d = ImageDraw.Draw(img_txt)
    sentence_list = []
    for i in range(captions.size(0)):
        cap = captions[i].detach().cpu().numpy()
        sentence = []
        for j, w in enumerate(cap):
            if w == 0:
                break
            word = ixtoword[int(w)].encode('ascii', 'ignore').decode('ascii')
            d.text(((j + off1) * (vis_size + off2), i * FONT_MAX), '%d:%s' % (j, word[:6]), font=fnt,
                   fill=(255, 255, 255, 255))
            sentence.append(word)
        sentence_list.append(sentence)
    return img_txt, sentence_list


def _to_uint8_range(imgs, vis_size):
    """(B,3,h,w) in [-1,1] -> (B,vis,vis,3) float in [0,255], bilinear."""
    imgs = F.interpolate(imgs.detach().float().cpu(), size=(vis_size, vis_size), mode='bilinear', align_corners=False)
    return ((imgs + 1) / 2 * 255).numpy().transpose(0, 2, 3, 1)


def _blend(img, one_map, vis_size, alpha):
    merged = Image.new('RGBA', (vis_size, vis_size), (0, 0, 0, 0))
    mask = Image.new('L', (vis_size, vis_size), (alpha))
    merged.paste(Image.fromarray(np.uint8(img)), (0, 0))
    merged.paste(Image.fromarray(np.uint8(one_map)), (0, 0), mask)
    return np.array(merged)[:, :, :3]


def build_super_images(real_imgs, captions, ixtoword, attn_maps, att_sze, lr_imgs=None, batch_size=None,
                       max_word_num=None, expand=pyramid_expand):
    """-> (uint8 image (nvis*(FONT_MAX + 2*vis), (T+2)*(vis+2), 3), sentences) or None when the caption strip and the
    map rows disagree in width.  The first 8 samples; per sample: caption strip / [low-res image | max-over-words map |
    one map per word] / [image | image blended with each map].  Maps are normalised with the min/max over the row."""
    batch_size = cfg.TRAIN.BATCH_SIZE if batch_size is None else batch_size
    max_word_num = cfg.TEXT.WORDS_NUM if max_word_num is None else max_word_num
    nvis = 8
    real_imgs = real_imgs[:nvis]
    if lr_imgs is not None:
        lr_imgs = lr_imgs[:nvis]
    vis_size = att_sze * 16 if att_sze == 17 else real_imgs.size(2)
    text_convas = np.ones([batch_size * FONT_MAX, (max_word_num + 2) * (vis_size + 2), 3], dtype=np.uint8)
    for i in range(max_word_num):
        text_convas[:, (i + 2) * (vis_size + 2):(i + 3) * (vis_size + 2), :] = COLOR_DIC[i]
    real = _to_uint8_range(real_imgs, vis_size)
    lr = _to_uint8_range(lr_imgs, vis_size) if lr_imgs is not None else None
    middle_pad = np.zeros([vis_size, 2, 3])
    post_pad = np.zeros([vis_size, vis_size, 3])
    text_map, sentences = drawCaption(text_convas, captions, ixtoword, vis_size)
    text_map = np.asarray(text_map).astype(np.uint8)
    up = vis_size // att_sze
    img_set = []
    for i in range(nvis):
        attn = attn_maps[i].detach().float().cpu().view(1, -1, att_sze, att_sze)
        attn = torch.cat([attn.max(dim=1, keepdim=True)[0], attn], 1).view(-1, 1, att_sze, att_sze)
        attn = attn.repeat(1, 3, 1, 1).numpy().transpose(0, 2, 3, 1)
        num_attn = attn.shape[0]
        img = real[i]
        row = [img if lr is None else lr[i], middle_pad]
        row_merge = [img, middle_pad]
        maps = [expand(attn[j], sigma=20, upscale=up) if up > 1 else attn[j] for j in range(num_attn)]
        lo = min([1] + [m.min() for m in maps])
        hi = max([0] + [m.max() for m in maps])
        for j in range(max_word_num + 1):
            if j < num_attn:
                one_map = (maps[j] - lo) / (hi - lo) * 255
                merged = _blend(img, one_map, vis_size, 210)
            else:
                one_map, merged = post_pad, post_pad
            row += [one_map, middle_pad]
            row_merge += [merged, middle_pad]
        row, row_merge = np.concatenate(row, 1), np.concatenate(row_merge, 1)
        txt = text_map[i * FONT_MAX:(i + 1) * FONT_MAX]
        if txt.shape[1] != row.shape[1]:
            print('txt', txt.shape, 'row', row.shape)
            return None
        img_set.append(np.concatenate([txt, row, row_merge], 0))
    return np.concatenate(img_set, 0).astype(np.uint8), sentences


def build_super_images2(real_imgs, captions, cap_lens, ixtoword, attn_maps, att_sze, vis_size=256, topK=5,
                        expand=pyramid_expand):
    """Top-K variant (utils.py:214-316): per sample the K words with the largest thresholded attention mass, each map
    thresholded at 2/len and normalised on its own; rows = caption strip / blended images."""
    batch_size = real_imgs.size(0)
    max_word_num = int(np.max(cap_lens))
    text_convas = np.ones([batch_size * FONT_MAX, max_word_num * (vis_size + 2), 3], dtype=np.uint8)
    real = _to_uint8_range(real_imgs, vis_size)
    middle_pad = np.zeros([vis_size, 2, 3])
    text_map, sentences = drawCaption(text_convas, captions, ixtoword, vis_size, off1=0)
    text_map = np.asarray(text_map).astype(np.uint8)
    up = vis_size // att_sze
    img_set = []
    for i in range(len(attn_maps)):
        attn = attn_maps[i].detach().float().cpu().view(-1, 1, att_sze, att_sze)
        attn = attn.repeat(1, 3, 1, 1).numpy().transpose(0, 2, 3, 1)
        num_attn = int(cap_lens[i])
        thresh = 2. / float(num_attn)
        img = real[i]
        row_merge, row_txt, conf_score = [], [], []
        for j in range(num_attn):
            one_map = attn[j]
            conf_score.append(np.sum(one_map * (one_map > (2. * thresh))))
            one_map = one_map * (one_map > thresh)
            if up > 1:
                one_map = expand(one_map, sigma=20, upscale=up)
            lo, hi = one_map.min(), one_map.max()
            one_map = (one_map - lo) / (hi - lo) * 255
            row_merge.append(np.concatenate([_blend(img, one_map, vis_size, 180), middle_pad], 1))
            row_txt.append(text_map[i * FONT_MAX:(i + 1) * FONT_MAX, j * (vis_size + 2):(j + 1) * (vis_size + 2), :])
        order = np.argsort(conf_score)[::-1][:topK]
        row_merge = np.concatenate([row_merge[k] for k in order], 1)
        txt = np.concatenate([row_txt[k] for k in order], 1)
        if txt.shape[1] != row_merge.shape[1]:
            print('Warnings: txt', txt.shape, 'row_merge', row_merge.shape)
            return None
        img_set.append(np.concatenate([txt, row_merge], 0))
    return np.concatenate(img_set, 0).astype(np.uint8), sentences
