"""Checkpoint helpers with the reference's names and on-disk formats
(util/checkpoint_util.py of the reference): an encoder-only ``checkpoint.tar``
holding ``{'params': encoder.state_dict()}`` and full-model checkpoints holding
``{'epoch','state_dict','optimizer','max_iou'}``.  Host glue only."""
import os

import torch


def load_pretrain_checkpoint(model, pretrain_checkpoint_path):
    if pretrain_checkpoint_path is None:
        raise ValueError('Pretrained checkpoint must be given.')
    print('Load encoder module from pretrained checkpoint...')
    blob = torch.load(os.path.join(pretrain_checkpoint_path, 'checkpoint.tar'), map_location='cpu')['params']
    own = model.state_dict()
    own.update({'encoder.' + k: v for k, v in blob.items() if 'encoder.' + k in own})
    model.load_state_dict(own)
    return model


def load_model_checkpoint(model, model_checkpoint_path, optimizer=None, mode='test'):
    try:
        ckpt = torch.load(model_checkpoint_path, map_location='cpu')
        start_iter, start_iou = ckpt['epoch'], ckpt['max_iou']
    except Exception:
        raise ValueError('Model checkpoint file must be correctly given (%s).' % model_checkpoint_path)
    model.load_state_dict(ckpt['state_dict'], strict=False)
    if mode == 'test':
        print('Load model checkpoint at Iteration %d (IoU %f)...' % (start_iter, start_iou))
        return model
    try:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    except Exception:
        print('Checkpoint does not include optimizer state dict...')
    print('Resume from checkpoint at Iteration %d (IoU %f)...' % (start_iter, start_iou))
    return model, optimizer


def save_pretrain_checkpoint(model, output_path):
    torch.save(dict(params=model.encoder.state_dict()), os.path.join(output_path, 'checkpoint.tar'))
