"""Caption-txt / img_id on-disk layout of the reference's image datasets (dataset/retrieval_dataset.py:30-135 train,
:140-260 eval), so real Multi30K / MSCOCO annotation directories can be used unchanged.

Layout under `root_dir` (config keys `root_dir`, `image_root`, `dataset`, `generated_caption_dir`, `generated_caption_type`,
`train_file`):
  img_id/train_id.txt | val_id.txt | test_id_2016.txt        one image id per line (its line number is the `idx` of the soft labels)
  img_id/image_ids.txt                                       mscoco only: "<id> <file name>" per line
  <generated_caption_dir>/<image id>.txt                     the generated caption of that image ('feats': <id>.npy, [*, 768])
  <ann file>                                                 "<cap_id> <caption>" per line; cap_id = "<image>[.jpg]#enc#<k>";
                                                             the i-th translated file names its captions "#enc2<lang>#"

This module is host-side text plumbing only: it yields image PATHS (JPEG decoding / augmentation stays with the caller, as
SURVEY marks it out of scope) plus exactly the strings / indices the reference's Dataset classes hand to the train loop.
Checked against those classes run on a generated store (tests/golden/caption_store.json).
"""
import os
import re

_PUNCT = re.compile(r"([,.'!?\"()*#:;~])")
_SPACES = re.compile(r"\s{2,}")


def pre_caption(caption, max_words):
    """dataset/utils.py:31-55: lower-case, punctuation -> space, '-' and '/' -> space, '<person>' -> 'person', runs of
    whitespace collapsed, at most `max_words` words; an empty result is an error."""
    raw = caption
    caption = _PUNCT.sub(' ', caption.lower()).replace('-', ' ').replace('/', ' ').replace('<person>', 'person')
    caption = _SPACES.sub(' ', caption).rstrip('\n').strip(' ')
    words = caption.split(' ')
    if len(words) > max_words:
        caption = ' '.join(words[:max_words])
    if not len(caption):
        raise ValueError(f'pre_caption yields invalid text (raw: {raw})')
    return caption


def get_video_id(cap_id):
    """dataset/retrieval_dataset.py:20-24: 'img.jpg#enc#0' -> 'img'."""
    vid = cap_id.split('#')[0]
    return vid[:-4] if vid.endswith('.jpg') or vid.endswith('.mp4') else vid


def _lines(path):
    with open(path) as f:
        return [ln.strip('\n') for ln in f.readlines()]


def _image_map(root_dir, sep_once):
    out = {}
    for ln in _lines(os.path.join(root_dir, 'img_id', 'image_ids.txt')):
        tmp = ln.split(' ', 1) if sep_once else ln.split(' ')
        out[tmp[0]] = tmp[1]
    return out


def _generated_captions(config, ids, image_map):
    """{image id: generated caption text} (or a [*, 768] float tensor for generated_caption_type == 'feats')."""
    out, gdir = {}, config['generated_caption_dir']
    for i in ids:
        if config['generated_caption_type'] == 'feats':
            import numpy as np
            import torch
            out[i] = torch.tensor(np.load(os.path.join(gdir, f'{i}.npy'), allow_pickle=True)).reshape(-1, 768)
            continue
        # mscoco: the reference strips the CHARACTERS '.', 'j', 'p', 'g' from both ends of the file name (str.strip), not the suffix
        name = image_map[i].strip('.jpg') if config['dataset'] == 'mscoco' else i
        with open(os.path.join(gdir, f'{name}.txt')) as f:
            out[i] = f.read().strip('\n')
    return out


class TrainAnnotations:
    """re_train_dataset_caption minus the image decode: len() = captions of the first (source-language) file;
    record(i) -> (image_path, [caption per language], generated caption, image index, cap_id of the last language)."""

    def __init__(self, config, max_words=30):
        self.config, self.max_words = config, max_words
        root = config['root_dir']
        self.image_root, self.dataset = config['image_root'], config['dataset']
        self.image_map = _image_map(root, sep_once=True) if self.dataset == 'mscoco' else None
        ids = _lines(os.path.join(root, 'img_id', 'train_id.txt'))
        self.generated_caption_dict = _generated_captions(config, ids, self.image_map)
        self.captions, self.cap_ids, self.languages = {}, [], []
        for i, ann in enumerate(config['train_file']):
            if i != 0:
                self.languages.append(ann.rsplit('/', 1)[-1].split('.', 1)[0].split('2', 1)[-1])
            cmap = {}
            with open(os.path.join(root, ann)) as f:
                for ln in f.readlines():
                    cap_id, caption = ln.strip().split(' ', 1)
                    cmap[cap_id] = caption
                    if i == 0:
                        self.cap_ids.append(cap_id)
            self.captions[i] = cmap
        self.img_ids = {i: n for n, i in enumerate(ids)}

    def __len__(self):
        return len(self.captions[0])

    def record(self, index):
        cap_id = self.cap_ids[index]
        image_id = get_video_id(cap_id)
        name = self.image_map[image_id] if self.dataset == 'mscoco' else f'{image_id}.jpg'
        captions = []
        for k, v in self.captions.items():
            if k != 0:      # as the reference: the replacement is applied to the running cap_id (a no-op after the first language)
                cap_id = cap_id.replace('#enc#', f'#enc2{self.languages[k - 1]}#')
            captions.append(pre_caption(v[cap_id], self.max_words))
        return f'{self.image_root}/{name}', captions, self.generated_caption_dict[image_id], self.img_ids[image_id], cap_id


class EvalAnnotations:
    """re_eval_dataset_caption minus the image decode: .text (pre-captioned), .image (file names in first-seen order),
    .txt2img / .img2txt (the ground truth itm_eval ranks against), .cap_ids, optional .text_trans;
    record(i) -> (image_path, generated caption, i)."""

    def __init__(self, config, ann_file, max_words=30, test_trans_file=None, type='eval'):
        self.config, self.max_words = config, max_words
        root = config['root_dir']
        self.image_root, self.dataset = config['image_root'], config['dataset']
        if self.dataset == 'mscoco':
            self.image_map = _image_map(root, sep_once=False)
            language = ann_file.split('.')[0].split('_')[-1]
            id_path = f'{language}_val_id.txt' if type == 'eval' else f'{language}_test_id.txt'
        else:
            self.image_map = None
            id_path = 'val_id.txt' if type == 'eval' else 'test_id_2016.txt'
        ids = _lines(os.path.join(root, 'img_id', id_path))
        self.generated_caption_dict = _generated_captions(config, ids, self.image_map)
        self.text, self.cap_ids, self.txt2img, self.img2txt, self.image = [], [], {}, {}, []
        seen = {}
        with open(os.path.join(root, ann_file)) as f:
            for txt_id, ln in enumerate(f.readlines()):
                cap_id, caption = ln.strip().split(' ', 1)
                image_id = get_video_id(cap_id)
                if image_id not in seen:
                    seen[image_id] = len(seen)
                    self.image.append(image_id + '.jpg')
                img_id = seen[image_id]
                self.text.append(pre_caption(caption, max_words))
                self.txt2img[txt_id] = img_id
                self.img2txt.setdefault(img_id, []).append(txt_id)
                self.cap_ids.append(cap_id)
        if test_trans_file is not None:
            self.text_trans = []
            with open(test_trans_file) as f:
                for ln in f.readlines():
                    self.text_trans.append(pre_caption(ln.strip().split(' ', 1)[1], max_words))

    def __len__(self):
        return len(self.image)

    def record(self, index):
        image_id = self.image[index]
        path = f'{self.image_root}/{image_id}'
        if self.dataset == 'mscoco':
            path = f"{self.image_root}/{self.image_map[image_id.strip('.jpg')]}"
        return path, self.generated_caption_dict[image_id.split('.')[0]], index
