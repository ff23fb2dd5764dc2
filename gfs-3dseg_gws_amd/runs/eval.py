"""GFS metric with the reference's signature (runs/eval.py:9-108 of the reference).

The reference walks every point in Python with ``int(tensor[j, k])`` - one device
sync per point.  Here the three per-class counters (ground truth, predicted,
true positive; all in class-NAME order via ``all_learning_order``) come from
``bincount`` on the device the labels live on, and - when torch.distributed is
initialised (block-sharded data parallel) - one all-reduce of 3 x n_cls int64.
"""
import numpy as np
import torch


def confusion_counts(pred_labels_list, gt_labels_list, all_learning_order, num_class):
    """-> int64 tensor (3, num_class): rows = gt count, predicted count, true positives."""
    dev = None
    for t in list(pred_labels_list) + list(gt_labels_list):
        if torch.is_tensor(t) and t.is_cuda:
            dev = t.device
            break
    dev = dev or torch.device("cpu")
    order = torch.as_tensor(np.asarray(all_learning_order), dtype=torch.int64, device=dev)
    counts = torch.zeros((3, num_class), dtype=torch.int64, device=dev)
    for pred, gt in zip(pred_labels_list, gt_labels_list):
        pred = torch.as_tensor(pred).to(dev).reshape(-1).long()
        gt = torch.as_tensor(gt).to(dev).reshape(-1).long()
        gt_name, pred_name = order[gt], order[pred]
        counts[0] += torch.bincount(gt_name, minlength=num_class)
        counts[1] += torch.bincount(pred_name, minlength=num_class)
        counts[2] += torch.bincount(gt_name[gt == pred], minlength=num_class)
    return counts


def evaluate_metric_GFS(logger, pred_labels_list, gt_labels_list, test_classes, novel_classes, all_learning_order,
                        scannet=False):
    """Returns (mean_iou, base_iou, novel_iou, hm, iou_array) like the reference; on
    ScanNet class name 0 is excluded from every mean."""
    assert len(pred_labels_list) == len(gt_labels_list)
    logger.cprint('*****Test Classes: {0}*****'.format(test_classes))
    num_class = len(test_classes)
    counts = confusion_counts(pred_labels_list, gt_labels_list, all_learning_order, num_class)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        # every rank passes ITS shard of the validation blocks; RCCL ("nccl") only moves device tensors
        home = counts.device
        if torch.distributed.get_backend() == "nccl" and not counts.is_cuda:
            counts = counts.to(torch.device("cuda", torch.cuda.current_device()))
        torch.distributed.all_reduce(counts)
        counts = counts.to(home)
    gt_c, pos_c, tp_c = (counts[i].cpu().numpy() for i in range(3))

    iou_list, base_ious, novel_ious = [], [], []
    for c in range(num_class):
        union = float(gt_c[c] + pos_c[c] - tp_c[c])
        if union == 0.0:           # the reference divides Python numbers here (runs/eval.py:52): same exception
            raise ZeroDivisionError("evaluate_metric_GFS: class %d appears in neither the labels nor the predictions" % c)
        iou = tp_c[c] / union
        logger.cprint('----- [class %d]  IoU: %f -----' % (c, iou))
        iou_list.append(iou)
        if scannet and c == 0:
            continue
        (novel_ious if c in novel_classes else base_ious).append(iou)
    if scannet:
        iou_list = iou_list[1:]
    mean_iou = np.array(iou_list).mean()
    base_iou = np.array(base_ious).mean()
    novel_iou = np.array(novel_ious).mean()
    hm = 2 * base_iou * novel_iou / (base_iou + novel_iou)
    logger.cprint('mean-iou: {}'.format(mean_iou))
    logger.cprint('base-iou: {}'.format(base_iou))
    logger.cprint('novel-iou: {}'.format(novel_iou))
    logger.cprint('hm-iou: {}'.format(hm))
    return mean_iou, base_iou, novel_iou, hm, np.array(iou_list)
