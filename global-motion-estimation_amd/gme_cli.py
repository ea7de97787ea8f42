#!/usr/bin/env python3
"""One command line for the scripts of the package (the authors' roadmap item 3,
``recap_future_updates.md:8,14``: "a CLI that shows the user the various possibilities for the
parameters and the usages of the various scripts").  EXTENSION: the reference has two separate
argparse scripts (bbme.py:658-714, results.py:117-138); their flags are kept as they are.

    python gme_cli.py bbme    -p <video|frame dir> -fi 13 -bs 16 -sw 16 -sp 0     # bbme.py main
    python gme_cli.py results -v <name under resources/videos> -f 1 [--model similarity] [--suggest]   # results.py main
    python gme_cli.py suggest -p <video|frame dir> [-fi 1] [-f 1]                 # parameter heuristics
    python gme_cli.py info                                                          # searches, norms, models, device
"""
import argparse
import sys


def _parser():
    ap = argparse.ArgumentParser(prog="gme_cli.py", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="command", required=True)
    b = sub.add_parser("bbme", help="motion field between two frames, plain and hierarchical (bbme.py:617-649)")
    b.add_argument("-p", "--video-path", dest="path", type=str, required=True, help="video file, frame directory, .npy or .y4m")
    b.add_argument("-fi", "--frame-index", dest="fi", type=int, required=True, help="index of the current frame (the previous one is fi - 3)")
    b.add_argument("-pn", "--p-norm", dest="pnorm", type=int, default=0, help="parsed and ignored, as upstream (the norm stays MSE)")
    b.add_argument("-bs", "--block-size", dest="block_size", type=int, default=12)
    b.add_argument("-sw", "--search-window", dest="search_window", type=int, default=8)
    b.add_argument("-sp", "--searching-procedure", dest="searching_procedure", type=int, default=1,
                   help="0: exhaustive, 1: three-step, 2: 2-D log, 3: diamond")
    r = sub.add_parser("results", help="global motion estimation + compensation + PSNR over a whole video (results.py:14-138)")
    r.add_argument("-v", "--video-name", dest="path", type=str, required=True, help="name under resources/videos (file or frame directory)")
    r.add_argument("-f", "--frame-distance", dest="fd", type=str, required=False)
    r.add_argument("--block-size", type=int, default=None, help="motion.BBME_BLOCK_SIZE for this run (the authors patched the constant by hand)")
    r.add_argument("--outlier-fraction", type=float, default=None, help="motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE for this run")
    r.add_argument("--model", choices=("affine", "translation", "similarity"), default="affine",
                   help="motion model fitted per level (roadmap.solve_model); affine is the reference")
    r.add_argument("--suggest", action="store_true",
                   help="pick block size and outlier fraction for this video with roadmap.suggest_parameters (middle pair); "
                        "explicit --block-size / --outlier-fraction win")
    s = sub.add_parser("suggest", help="heuristic block size / search window / outlier fraction for a frame pair (roadmap.suggest_parameters)")
    s.add_argument("-p", "--video-path", dest="path", type=str, required=True)
    s.add_argument("-fi", "--frame-index", dest="fi", type=int, default=1)
    s.add_argument("-f", "--frame-distance", dest="fd", type=int, default=1)
    sub.add_parser("info", help="list searches, norms, motion models and the device")
    return ap


def main(argv=None):
    args = _parser().parse_args(argv)
    if args.command == "bbme":
        import bbme
        return bbme.main(args)
    if args.command == "results":
        import motion
        import results
        old = motion.BBME_BLOCK_SIZE, motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE
        try:
            if args.suggest:
                import os
                import roadmap
                import utils
                frames = utils.get_video_frames(os.path.join("resources", "videos", args.path))
                fd = int(args.fd) if args.fd is not None else results.FRAME_DISTANCE
                mid = max(fd, len(frames) // 2)
                hint = roadmap.suggest_parameters(frames[mid - fd], frames[mid])
                print("suggested for this video: {}".format(hint))
                motion.BBME_BLOCK_SIZE = hint["block_size"]
                motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE = hint["outlier_fraction"]
            if args.block_size is not None:
                motion.BBME_BLOCK_SIZE = args.block_size
            if args.outlier_fraction is not None:
                motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE = args.outlier_fraction
            return results.main(args)
        finally:
            motion.BBME_BLOCK_SIZE, motion.MOTION_VECTOR_ERROR_THRESHOLD_PERCENTAGE = old
    if args.command == "suggest":
        import roadmap
        import utils
        frames = utils.get_video_frames(args.path)
        out = roadmap.suggest_parameters(frames[args.fi - args.fd], frames[args.fi])
        print("frame shape: {}".format(frames[0].shape))
        for k, v in out.items():
            print("{}: {}".format(k, v))
        return out
    import _gme_native
    import roadmap
    print("searching procedures (-sp): 0 exhaustive, 1 three-step, 2 2-D log, 3 diamond   (bbme.py:609-614)")
    print("norms: 0 MAE, 1 MSE (bbme.py:608; the bbme script always uses MSE, as upstream)")
    print("motion models (roadmap.global_motion_estimation): " + ", ".join(roadmap.MODELS))
    try:
        print("device: " + _gme_native.default_context().info()["name"])
    except Exception as e:      # noqa: BLE001 -- no GPU: say so, the listing above is still useful
        print("device: none (%s)" % e)
    return None


if __name__ == "__main__":
    main(sys.argv[1:])
