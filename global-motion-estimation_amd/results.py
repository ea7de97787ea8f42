"""Whole-video driver -- the reference's ``results.py`` (lines 14-138) on the GPU path.

Same outputs (five PNG sets and ``psnr_records.json`` with ``str(complex)`` values) and the same
command line (``-v`` video name under ``resources/videos``, ``-f`` frame distance), but the
video's pairs stream through the device in chunks (``sequence.estimate_stream``: chunk k + 1 is uploaded
while chunk k is estimated and compensated) instead of one pair at a time, and OpenCV is optional: frames may
come from an image directory / ``.npy`` / ``.y4m`` (``utils.get_video_frames``) and PNGs are
written with PIL when cv2 is missing.
"""
import argparse
import os
import shutil
from json import dump

import numpy as np

import motion
from sequence import estimate_stream
from utils import draw_motion_field, get_video_frames, write_image

FRAME_DISTANCE = 1          # results.py:11
STREAMS = 2                 # lanes (context + HIP stream + chunk-sized device sequence) the chunks of a video rotate through
CHUNK_PAIRS = 512           # most pairs per chunk (a lane's device sequence holds that many + fd frames); StreamEstimator.schedule shrinks the last ones


def process_frames(frames, frame_distance=FRAME_DISTANCE, save_path=None, progress=False, model="affine"):
    """results.py:41-112 for a list of grayscale frames -> ``{str(idx): str(psnr)}``.

    ``model`` other than "affine" (the reference) selects one of roadmap.MODELS -- an extension, see roadmap.py.

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
    # the video stays in host memory and streams through a few lanes: one lane's upload runs beside the other lanes'
    # kernels, one host thread drives all of them (split-phase calls) and does their 3x3 solves in between; every device
    # object is released before this returns
    solve = None
    if model != "affine":
        import roadmap
        solve = lambda sums: roadmap.solve_model(sums, model)       # noqa: E731
    field_shape = (int(shape[0] / bs), int(shape[1] / bs), 2)
    written = {}

    def write_chunk(p0, p1, comp, params, psnr):
        """The image sets of one finished chunk (results.py:62-106) and the records so far (results.py:109-112).  The chunk's
        compensated frames come in one read and memory stays bounded by the chunk (a whole-video host array of
        compensated frames -- 4 GB for 2000 frames of 1080p -- and one blocking read per pair used to stall the host thread
        that drives all lanes)."""
        for k in range(p1 - p0):
            p, idx = p0 + k, p0 + k + fd
            previous, current, compensated = frames[p], frames[idx], comp[k]
            model_motion_field = motion.get_motion_field_affine(field_shape, parameters=params[k])
            write_image(os.path.join(save_path, "frames", "") + str(idx - 5) + ".png", previous)
            write_image(os.path.join(save_path, "compensated", "") + str(idx - 5) + ".png", compensated)
            diff_curr_prev = np.absolute(current.astype("int") - previous.astype("int")).astype("uint8")
            diff_curr_comp = np.absolute(current.astype("int") - compensated.astype("int")).astype("uint8")
            write_image(os.path.join(save_path, "curr_prev_diff", "") + str(idx) + ".png", diff_curr_prev)
            write_image(os.path.join(save_path, "curr_comp_diff", "") + str(idx) + ".png", diff_curr_comp)
            write_image(os.path.join(save_path, "model_motion_field", "") + str(idx) + ".png",
                        draw_motion_field(previous, model_motion_field))
            written[idx] = str(-1 if psnr[k] == -1 else complex(psnr[k], 0.0))
        with open(save_path + "psnr_records.json", "w") as outfile:        # chunks of different lanes may finish out of order
            dump({str(i): written[i] for i in sorted(written)}, outfile)

    # the video stays in host memory and streams through a few lanes: one lane's upload runs beside the other lanes'
    # kernels, one host thread drives all of them (split-phase calls) and does their 3x3 solves in between; every device
    # object is released before this returns
    params, psnr = estimate_stream(frames, fd, chunk_pairs=CHUNK_PAIRS, streams=STREAMS, solve=solve,
                                   on_chunk=write_chunk if save_path is not None else None)
    for idx in range(fd, len(frames)):
        p = idx - fd
        if progress:
            j = (idx + 1) / len(frames)
            print("[%-20s] %d/%d frames" % ("=" * int(20 * j), idx, len(frames)))
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
    return process_frames(frames, frame_distance, save_path, progress=True, model=getattr(args, "model", None) or "affine")


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Launches GME and yields results")
    parser.add_argument("-v", "--video-name", dest="path", type=str, required=True,
                        help="name of the video to analyze (file or frame directory under resources/videos)")
    parser.add_argument("-f", "--frame-distance", dest="fd", type=str, required=False, help="frame displacement")
    main(parser.parse_args())
