"""Final detection post-processing on the device (interface of the reference's yolox/models/post_process.py:9-85).

``postprocess`` keeps the reference's signature (lists of per-frame tensors) and runs tscd_amd's NMS kernels:
candidate expansion (+ decode_reg_preds5 when ``reg_deltas`` are given), an LDS bitonic sort per frame, per-class
greedy suppression and an ordered compaction.  One D2H of the per-frame counts at the end sizes the ragged outputs.
"""
import torch

from tscd_amd import ops

# Reference-faithful switch: the reference's final NMS is torchvision.ops.batched_nms on the evaluator's fp16 tensors
# (post_process.py:58-63,73-78), whose coordinate-offset trick quantises the boxes of the high class ids to 4-8 px.  The product
# default (0) suppresses exactly per class in fp32 (bounded in tests/test_deviation_bounds.py);
# COORD_TRICK_MAX_BOXES = n applies torchvision's arithmetic to frames with at most n candidates -- 5000 is torchvision's own rule on
# a GPU (above it batched_nms runs per class without offsets, which is what the default does everywhere).
COORD_TRICK_MAX_BOXES = int(__import__("os").environ.get("TSCD_NMS_COORD_TRICK", "0"))


def nms_frames(det_rows, frame_counts, num_classes, cls_logits=None, obj_logits=None, reg_deltas=None, conf_thre=0.001,
               nms_thre=0.5, return_idx=False, coord_trick=None):
    """det_rows [N, 7+NC] fp16 (pred_result rows of the local frames, frame-major), frame_counts: python list.
    Returns (result, result_ori[, keep]) lists with [k,7] tensors or None, like the reference."""
    dev = det_rows.device
    L = len(frame_counts)
    off = [0]
    for n in frame_counts:
        off.append(off[-1] + n)
    frame_off = ops.h2d(off, torch.int32, dev)
    nmax = max(frame_counts) if frame_counts else 0
    if nmax == 0:
        return [None] * L, [None] * L
    # every (proposal, class) pair can pass the score filter: the candidate arrays are sized for all of them (no sync to count the
    # survivors; the sort and the suppression only touch what survived).  The ImageNet-VID exp sets no upper proposal clamp
    # (exps/TSCD_VID/vid_tscd_large.py:42): up to A anchors x 30 classes = 204 120 candidates per frame at 576^2 -- 7 MB of workspace each
    cap0 = nmax * num_classes
    det_rows = det_rows.contiguous()
    half = lambda t: t.to(torch.float16).contiguous()
    trick = COORD_TRICK_MAX_BOXES if coord_trick is None else int(coord_trick)
    rows0, idx0, cnt0 = ops.batched_nms(det_rows, frame_off, L, num_classes, conf_thre, nms_thre, cap0, 0,
                                        half(cls_logits), half(obj_logits).reshape(-1), half(reg_deltas), max_rows=nmax, coord_trick=trick)
    rows1, idx1, cnt1 = ops.batched_nms(det_rows, frame_off, L, num_classes, conf_thre, nms_thre, max(nmax, 1), 1, max_rows=nmax,
                                        coord_trick=trick)
    counts = torch.cat((cnt0, cnt1)).cpu().tolist()      # the one sync of the eval tail
    res, ori, keep = [], [], []
    for i in range(L):
        k0, k1 = counts[i], counts[L + i]
        # the reference leaves output[i] / output_ori[i] = None when no refined detection survives the threshold
        res.append(rows0[i, :k0] if k0 > 0 else None)
        ori.append(rows1[i, :k1] if k0 > 0 and k1 > 0 else None)
        keep.append(idx0[i, :k0])
    if return_idx:
        return res, ori, keep
    return res, ori


def postprocess(prediction, num_classes, fc_outputs, conf_output, reg_output, conf_thre=0.001, nms_thre=0.5, cls_sig=True,
                return_idx=False):
    """Reference signature: prediction list of [n_i,7+NC]; fc_outputs list of [n_i,NC] logits; conf_output list of [n_i]
    logits; reg_output list of [n_i,4] ALREADY-decoded xyxy boxes (the reference decodes before calling).  Because the
    fused kernel decodes deltas itself, already-decoded boxes are passed through by writing them into the detection rows
    with zero deltas."""
    if not cls_sig:
        raise NotImplementedError("TSCD calls postprocess with cls_sig=True")
    counts = [0 if p is None else p.shape[0] for p in prediction]
    live = [i for i, n in enumerate(counts) if n > 0]
    if not live:
        return [None] * len(prediction), [None] * len(prediction)
    det = torch.cat([prediction[i] for i in live], 0).to(torch.float16)
    cls = torch.cat([fc_outputs[i] for i in live], 0)
    obj = torch.cat([conf_output[i].reshape(-1) for i in live], 0)
    if reg_output is not None:
        # NOTE: 'original' detections use the un-refined boxes, so keep them aside for the second NMS
        det_ref = det.clone()
        det_ref[:, :4] = torch.cat([reg_output[i] for i in live], 0).to(torch.float16)
    else:
        det_ref = det
    zeros = torch.zeros((det.shape[0], 4), dtype=torch.float16, device=det.device)
    r0, _ = nms_frames(det_ref, [counts[i] for i in live], num_classes, cls, obj, zeros, conf_thre, nms_thre)
    _, r1 = nms_frames(det, [counts[i] for i in live], num_classes, cls, obj, zeros, conf_thre, nms_thre)
    out, out_ori = [None] * len(prediction), [None] * len(prediction)
    for j, i in enumerate(live):
        out[i], out_ori[i] = r0[j], r1[j]
    return out, out_ori
