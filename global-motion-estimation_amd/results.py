"""Whole-video driver -- the reference's ``results.py`` (lines 14-138) on the GPU path.

Same outputs (five PNG sets and ``psnr_records.json`` with ``str(complex)`` values) and the same
command line (``-v`` video name under ``resources/videos``, ``-f`` frame distance), but the
video's pairs are estimated and compensated in one device-resident batch
(``sequence.ShardedSequence``) instead of one pair at a time, and OpenCV is optional: frames may
come from an image directory / ``.npy`` / ``.y4m`` (``utils.get_video_frames``) and PNGs are
written with PIL when cv2 is missing.
"""
import argparse
import os
import shutil
from json import dump

import numpy as np

import motion
from sequence import ShardedSequence
from utils import draw_motion_field, get_video_frames, write_image

FRAME_DISTANCE = 1          # results.py:11
STREAMS = 3                 # HIP streams (pair ranges) per GPU: uploads run one host thread per range, the estimate one thread over all


def process_frames(frames, frame_distance=FRAME_DISTANCE, save_path=None, progress=False):
    """results.py:41-112 for a list of grayscale frames -> ``{str(idx): str(psnr)}``.

    With ``save_path`` the five image sets are written with the reference's (quirky) names:
    ``frames/`` and ``compensated/`` use ``idx-5``, the others ``idx`` (results.py:64-106).
    """
    frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    try:
        shape = frames[0].shape
    except Exception:
        raise Exception("Error reading video file: check the name of the video!")     # results.py:37-40
    fd = int(frame_distance)
    bs = int(motion.BBME_BLOCK_SIZE)
    psnr_dict = {}
    if len(frames) <= fd:
        return psnr_dict
    # a few streams per GPU: each lane uploads its range of the video in one copy and then works on it,
    # so one lane's host->device copy runs beside the other lanes' kernels; the estimate is driven by one host
    # thread over all streams (split-phase calls), whose 3x3 solves for one lane run beside the others' searches
    seq = ShardedSequence(shape[0], shape[1], len(frames), fd, streams=STREAMS, interleave=True)
    seq.load(frames)
    params = seq.estimate()                                   # motion.global_motion_estimation per pair
    psnr = seq.compensate(params)                             # compensate_frame + PSNR per pair
    field_shape = (int(shape[0] / bs), int(shape[1] / bs), 2)
    for idx in range(fd, len(frames)):
        p = idx - fd
        if progress:
            j = (idx + 1) / len(frames)
            print("[%-20s] %d/%d frames" % ("=" * int(20 * j), idx, len(frames)))
        if save_path is not None:
            previous, current = frames[p], frames[idx]
            compensated = seq.read_compensated(p)
            model_motion_field = motion.get_motion_field_affine(field_shape, parameters=params[p])
            write_image(os.path.join(save_path, "frames", "") + str(idx - 5) + ".png", previous)
            write_image(os.path.join(save_path, "compensated", "") + str(idx - 5) + ".png", compensated)
            diff_curr_prev = np.absolute(current.astype("int") - previous.astype("int")).astype("uint8")
            diff_curr_comp = np.absolute(current.astype("int") - compensated.astype("int")).astype("uint8")
            write_image(os.path.join(save_path, "curr_prev_diff", "") + str(idx) + ".png", diff_curr_prev)
            write_image(os.path.join(save_path, "curr_comp_diff", "") + str(idx) + ".png", diff_curr_comp)
            write_image(os.path.join(save_path, "model_motion_field", "") + str(idx) + ".png",
                        draw_motion_field(previous, model_motion_field))
        value = psnr[p]
        psnr_dict[str(idx)] = str(-1 if value == -1 else complex(value, 0.0))       # utils.PSNR returns cmath complex
        if save_path is not None:
            with open(save_path + "psnr_records.json", "w") as outfile:
                dump(psnr_dict, outfile)
    return psnr_dict


def main(args):
    video = args.path
    frame_distance = int(args.fd) if args.fd is not None else FRAME_DISTANCE
    video_path = os.path.join("resources", "videos", video)
    results_path = os.path.join("results", "")
    save_path = os.path.join(results_path, os.path.splitext(video)[0], "")
    if os.path.isdir(save_path):
        shutil.rmtree(save_path)
    os.makedirs(save_path)
    for sub in ("frames", "compensated", "curr_prev_diff", "model_motion_field", "curr_comp_diff"):
        os.mkdir(os.path.join(save_path, sub, ""))
    frames = get_video_frames(video_path)
    try:
        print("frame shape: {}".format(frames[0].shape))
    except Exception:
        raise Exception("Error reading video file: check the name of the video!")
    return process_frames(frames, frame_distance, save_path, progress=True)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Launches GME and yields results")
    parser.add_argument("-v", "--video-name", dest="path", type=str, required=True,
                        help="name of the video to analyze (file or frame directory under resources/videos)")
    parser.add_argument("-f", "--frame-distance", dest="fd", type=str, required=False, help="frame displacement")
    main(parser.parse_args())
