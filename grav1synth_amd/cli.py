"""`python -m grav1synth_amd diff SOURCE DENOISED -o OUT [-y] [-f FILTERS]` -- the front door of the path.

The `diff` command of the reference (Commands::Diff, /root/reference/src/main.rs:347-533, arguments :846-870) for .y4m
inputs, around g1s_diff_y4m_files_filtered: the same refusals in the same order, with the same texts, and like the
reference every refusal is a logged line and a normal exit --

  * an input path equal to the output path              (src/main.rs:354-360)
  * source path equal to denoised path                  (src/main.rs:362-368)
  * a filter chain that does not parse                  (src/main.rs:370-380: "Invalid filter chain: {e}")
  * an existing output without -y and without a "yes"   (src/main.rs:382-394)

-- then the frame-pair loop, finish, the table (src/main.rs:414-529) and the two closing lines (:531-532)."""
from __future__ import annotations

import argparse
import logging
import os
import sys
from typing import List, Optional

log = logging.getLogger("grav1synth")

SAME_AS_OUTPUT = ("Input and output paths are the same. This is probably a typo, because this would overwrite your "
                  "input. Exiting.")
SAME_INPUTS = ("Source and denoised paths are the same. This is probably a typo, because this would always compute an "
               "empty diff. Exiting.")
NOT_OVERWRITING = "Not overwriting existing file. Exiting."


def _confirm(prompt: str) -> bool:
    """dialoguer::Confirm::interact()?: y / n on the terminal.  Without a terminal dialoguer returns an error, which the
    reference's `?` turns into a non-zero exit (src/main.rs:382-394): the same here (main() reports it and returns 1)."""
    if not sys.stdin.isatty():
        raise OSError("IO error: not a terminal")
    try:
        return input(f"{prompt} [y/n] ").strip().lower() in ("y", "yes")
    except EOFError:
        raise OSError("IO error: unexpected end of input")


def _same_path(a: str, b: str) -> bool:
    """`PathBuf == PathBuf` as the reference compares its arguments (src/main.rs:354, :362): component by component, so
    repeated separators, a trailing separator and `.` components inside the path do not matter (a leading `./` and `..` do:
    std::path::Path::components)."""
    def components(p: str):
        parts = p.split("/")
        out = ["/"] if p.startswith("/") else []
        for i, c in enumerate(parts):
            if c == "" or (c == "." and (i > 0 or p.startswith("/"))):
                continue
            out.append(c)
        return out
    return components(a) == components(b)


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="grav1synth_amd", description="MI355X-native `grav1synth diff`")
    sub = ap.add_subparsers(dest="command", required=True)
    d = sub.add_parser("diff", help="Compares a source video to a denoised video and generates a film grain table (y4m inputs).")
    d.add_argument("source", help="The untouched source file to inspect.")
    d.add_argument("denoised", help="The denoised file to inspect.")
    d.add_argument("-o", "--output", required=True, help="The path to the output film grain table.")
    d.add_argument("-y", "--overwrite", action="store_true", help="Overwrite the output file without prompting.")
    d.add_argument("-f", "--filters", default=None,
                   help='A semicolon-separated list of filters to apply to the source before running the diff, e.g. '
                        '"crop:top=42,left=64".  crop: top, bottom, left, right.  resize: width, height, alg (hermite, catmullrom, '
                        "mitchell, lanczos, spline36) runs on the device; its arithmetic restates the video-resize crate, which "
                        "is not in the reference tree: the resized planes are UNVERIFIED against grav1synth's (a warning is logged).")
    d.add_argument("--device", type=int, default=-1, help="HIP device ordinal (default: the current device)")
    d.add_argument("--gpus", type=int, default=1,
                   help="shard the frames over the first N devices of the node (one generator each, batches dealt round-robin, "
                        "ordered merge on the host: the same table as one device).  From .y4m files the command is bound by "
                        "reading them (about 25 GB/s, 560 4K 10-bit frames a second) long before a second device matters; "
                        "does not combine with a resize filter or with --device")
    d.add_argument("--devices", default=None, help="the same with an explicit list of HIP ordinals, e.g. 0,2,3")
    e = sub.add_parser("estimate", help="Estimates the amount of noise in a source video, frame by frame (y4m input; the reference's "
                                        "`estimate`, feature \"unstable\").")
    e.add_argument("source", help="The source file to inspect.")
    e.add_argument("-o", "--output", required=True, help="The path to the output file.")
    e.add_argument("-y", "--overwrite", action="store_true", help="Overwrite the output file without prompting.")
    e.add_argument("--device", type=int, default=-1, help="HIP device ordinal (default: the current device)")
    return ap


def diff_command(source: str, denoised: str, output: str, overwrite: bool = False, filters: Optional[str] = None,
                 device: int = -1, confirm=_confirm, devices: Optional[List[int]] = None) -> int:
    """Returns the number of frame pairs diffed, or -1 when the command refused to run (a logged line, exit 0)."""
    from .filters import FilterChain, FilterError, Resize
    from .ingest import diff_y4m_files

    if _same_path(source, output) or _same_path(denoised, output):
        log.error(SAME_AS_OUTPUT)
        return -1
    if _same_path(source, denoised):
        log.error(SAME_INPUTS)
        return -1
    if filters is not None:
        try:
            fc = FilterChain(filters)
        except FilterError as e:
            log.error("Invalid filter chain: %s", e)
            return -1
        resizes = any(isinstance(f, Resize) for f in fc.filters)
        fc.close()
        if resizes and devices is not None and len(devices) > 1:
            log.error("A resize filter does not combine with --gpus / --devices (one chain, one device)")
            return -1
        if resizes and devices is not None:  # (`--devices 2`: one device -- the plain command on it, not the sharded one)
            device, devices = devices[0], None
        if resizes:
            log.warning("resize: the resampling arithmetic restates the video-resize crate (not in the reference tree): the "
                        "resized source, and the table made from it, are UNVERIFIED against grav1synth's")
    if os.path.exists(output) and not overwrite and not confirm(f"File {output} exists. Overwrite?"):
        log.warning(NOT_OVERWRITING)
        return -1
    frames, _unequal = diff_y4m_files(source, denoised, output, device=device, filters=filters, devices=devices)  # (logs "Computed diff for N frames")
    log.info("Done, wrote output file to %s", output)
    return frames


def estimate_command(source: str, output: str, overwrite: bool = False, device: int = -1, confirm=_confirm) -> int:
    """Commands::Estimate (src/main.rs:534-608): the refusals of :541-560, one estimate_plane_noise per frame, "filmgrn1" and a
    "{:.3}" line per frame (-1 for None), "Done, wrote output file to ...".  Returns the frame count, -1 after a refusal."""
    from .estimate import estimate_y4m_file

    if _same_path(source, output):
        log.error(SAME_AS_OUTPUT)
        return -1
    if os.path.exists(output) and not overwrite and not confirm(f"File {output} exists. Overwrite?"):
        log.warning(NOT_OVERWRITING)
        return -1
    frames = estimate_y4m_file(source, output, device=device)
    log.info("Done, wrote output file to %s", output)
    return frames


def main(argv: Optional[List[str]] = None) -> int:
    # a hardware queue per stream of the generator (the HIP runtime reads this when it starts: before the first GPU call)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(levelname)s %(message)s", stream=sys.stderr)
    if args.command == "diff":
        try:
            devices = None
            if args.gpus < 1:
                raise ValueError("--gpus: at least 1")
            if (args.devices or args.gpus > 1) and args.device >= 0:
                raise ValueError("--device does not combine with --gpus / --devices")
            if args.devices:
                devices = [int(x) for x in args.devices.split(",") if x.strip() != ""]
            elif args.gpus > 1:
                devices = list(range(args.gpus))
            diff_command(args.source, args.denoised, args.output, args.overwrite, args.filters, args.device, devices=devices)
        except Exception as e:  # `?` out of main: the error, a non-zero exit
            log.error("%s", e)
            return 1
    elif args.command == "estimate":
        try:
            estimate_command(args.source, args.output, args.overwrite, args.device)
        except Exception as e:
            log.error("%s", e)
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
