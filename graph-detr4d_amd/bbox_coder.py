"""NMSFreeCoder: the box decoding step right after the decoder.

Mirror of projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py:17-118 (same constructor arguments, same
`decode(preds_dicts)` contract and output dicts) with decode_single's ~15 torch ops (sigmoid, topk, index gathers,
denormalize_bbox's cat, range masks) run as one launch of gd4d_nms_free_decode_fwd for all batch elements; only the
final boolean compaction - whose size the host must learn - is left to torch, as in the reference.
"""

from . import ops
from .registry import BBOX_CODERS


@BBOX_CODERS.register_module()
class NMSFreeCoder:
    def __init__(self, pc_range, voxel_size=None, post_center_range=None, max_num=100, score_threshold=None,
                 num_classes=10):
        self.pc_range = pc_range
        self.voxel_size = voxel_size
        self.post_center_range = post_center_range
        self.max_num = max_num
        self.score_threshold = score_threshold
        self.num_classes = num_classes

    def encode(self):
        pass

    def _decode_batch(self, cls_scores, bbox_preds):
        """cls_scores (B, Q, num_classes) logits, bbox_preds (B, Q, code) -> list of per-sample dicts."""
        if self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only '
                                      'support post_center_range is not None for now!')       # nms_free_coder.py:92-95
        if cls_scores.shape[-1] != self.num_classes:
            raise ValueError(f'cls_scores has {cls_scores.shape[-1]} classes, coder expects {self.num_classes}')
        # `if self.score_threshold:` in the reference (:81): a threshold of 0 / None is not applied
        thr = self.score_threshold if self.score_threshold else None
        boxes, scores, labels, keep = ops.nms_free_decode_fwd(cls_scores.contiguous().float(),
                                                              bbox_preds.contiguous().float(),
                                                              self.post_center_range, self.max_num, thr)
        out = []
        for b in range(cls_scores.shape[0]):
            m = keep[b]
            out.append({'bboxes': boxes[b][m], 'scores': scores[b][m], 'labels': labels[b][m].long()})
        return out

    def decode_single(self, cls_scores, bbox_preds):
        """cls_scores (Q, num_classes), bbox_preds (Q, code) -> dict(bboxes, scores, labels)  (:47-96)."""
        return self._decode_batch(cls_scores[None], bbox_preds[None])[0]

    def decode(self, preds_dicts):
        """preds_dicts['all_cls_scores'] (num_layers, B, Q, num_classes), ['all_bbox_preds'] (num_layers, B, Q, code):
        the last decoder layer is decoded (:98-117)."""
        return self._decode_batch(preds_dicts['all_cls_scores'][-1], preds_dicts['all_bbox_preds'][-1])
