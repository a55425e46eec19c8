"""Proposal-list text files: the input format of the reference's trainer and tester
(/root/reference/ops/io.py:7-59; written by gen_proposal_list.py, read by ssn_dataset.py:158).

Two flavours share one layout::

    # <index>
    <video id or frame folder>
    <a>                 normalised list: a = 1           processed list: a = frame count
    <b>                 normalised list: b = 1           processed list: b = 1          (n_frame = int(a * b))
    <number of ground-truth instances>
    <label> <start> <end>                                   one line each
    <number of proposals>
    <label> <best IoU> <overlap with itself> <start> <end>  one line each

``load_proposal_file`` returns the reference's tuples ``(vid, n_frame, gt_boxes, pr_boxes)`` with the box fields
still as strings; ``process_proposal_list`` converts normalised positions to frame indices.
"""


def _records(lines):
    """Split the file into per-video records: runs of lines between '#' header lines."""
    rec = []
    for line in lines:
        if line.startswith('#'):
            if rec:
                yield rec
            rec = []
        else:
            rec.append(line.strip())
    if rec:
        yield rec


def load_proposal_file(filename):
    """-> [(vid, n_frame, [[label, start, end], ...], [[label, iou, overlap_self, start, end], ...]), ...]"""
    with open(filename) as f:
        lines = list(f)
    out = []
    for info in _records(lines):
        vid = info[0]
        n_frame = int(float(info[1]) * float(info[2]))
        n_gt = int(info[3])
        gt_boxes = [x.split() for x in info[4:4 + n_gt]]
        n_pr = int(info[4 + n_gt])
        pr_boxes = [x.split() for x in info[5 + n_gt:5 + n_gt + n_pr]]
        out.append((vid, n_frame, gt_boxes, pr_boxes))
    return out


def format_processed_record(idx, frame_path, frame_cnt, gt, prop):
    """One record of a processed list (gt: [label, start, end] ints; prop: [label, iou, overlap_self, start, end])."""
    text = "# {}\n{}\n{}\n1\n{}\n".format(idx, frame_path, frame_cnt, len(gt))
    text += "".join("{} {:d} {:d}\n".format(*x) for x in gt)
    text += "{}\n".format(len(prop))
    text += "".join("{} {:.04f} {:.04f} {:d} {:d}\n".format(*x) for x in prop)
    return text


def process_proposal_list(norm_proposal_list, out_list_name, frame_dict):
    """Normalised list + ``frame_dict[vid] = (frame_path, frame_cnt, ...)`` -> processed list on disk."""
    records = []
    for idx, (vid, _, gt_boxes, pr_boxes) in enumerate(load_proposal_file(norm_proposal_list)):
        frame_path, frame_cnt = frame_dict[vid][0], frame_dict[vid][1]
        gt = [[int(x[0]), int(float(x[1]) * frame_cnt), int(float(x[2]) * frame_cnt)] for x in gt_boxes]
        prop = [[int(x[0]), float(x[1]), float(x[2]), int(float(x[3]) * frame_cnt), int(float(x[4]) * frame_cnt)]
                for x in pr_boxes]
        records.append(format_processed_record(idx, frame_path, frame_cnt, gt, prop))
    with open(out_list_name, 'w') as f:
        f.writelines(records)
