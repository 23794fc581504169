"""Final detection post-processing on the device (interface of the reference's yolox/models/post_process.py:9-85).

``postprocess`` keeps the reference's signature (lists of per-frame tensors) and runs tscd_amd's NMS kernels:
candidate expansion (+ decode_reg_preds5 when ``reg_deltas`` are given), an LDS bitonic sort per frame, per-class
greedy suppression and an ordered compaction.  One D2H of the per-frame counts at the end sizes the ragged outputs.
"""
import torch

from tscd_amd import ops

MAX_CANDIDATES = 65536        # 16-bit candidate index in nms.hip's sort key (frames above 16384 candidates sort through global memory)


def nms_frames(det_rows, frame_counts, num_classes, cls_logits=None, obj_logits=None, reg_deltas=None, conf_thre=0.001,
               nms_thre=0.5, return_idx=False):
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
    cap0 = nmax * num_classes
    if cap0 > MAX_CANDIDATES:
        raise NotImplementedError(f"{nmax} proposals x {num_classes} classes exceeds the {MAX_CANDIDATES} candidates per frame the NMS "
                                  "sort can index (ImageNet-VID exp without maximal_limit: set maximal_limit <= "
                                  f"{MAX_CANDIDATES // num_classes}; documented limit, DESIGN.md)")
    det_rows = det_rows.contiguous()
    half = lambda t: t.to(torch.float16).contiguous()
    rows0, idx0, cnt0, ovf0 = ops.batched_nms(det_rows, frame_off, L, num_classes, conf_thre, nms_thre, cap0, 0,
                                              half(cls_logits), half(obj_logits).reshape(-1), half(reg_deltas))
    rows1, idx1, cnt1, ovf1 = ops.batched_nms(det_rows, frame_off, L, num_classes, conf_thre, nms_thre, max(nmax, 1), 1)
    counts = torch.cat((cnt0, cnt1, ovf0, ovf1)).cpu().tolist()      # the one sync of the eval tail
    if counts[2 * L] or counts[2 * L + 1]:
        raise RuntimeError("NMS per-class capacity (2048 boxes of one class in one frame) exceeded")
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
