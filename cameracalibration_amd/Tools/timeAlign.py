"""Timestamp alignment of the four camera directories: the feeder that turns front/back/left/right image folders into
complete 4-camera frame sets for ``BevGenerator.batch()``.

Mirrors the behaviour and the public names of the reference's Tools/timeAlign.py (``align_time`` :18-72, ``TimeParser``
:74-88, the ``--front/--back/--left/--right/--usb_align_thresh`` options :4-10).  Pure host-side list logic -- no pixels,
no GPU.  tests/golden/time_align.json (written by the reference's own function) pins the behaviour.
"""
from __future__ import annotations

import argparse
import os

CAMERA_DIRS = ("front", "back", "left", "right")

parser = argparse.ArgumentParser(description="Time Align for Images")
for _cam in CAMERA_DIRS:
    parser.add_argument("--" + _cam, type=str, default="./data/" + _cam)
parser.add_argument("--usb_align_thresh", type=float, default=0.1)
args = parser.parse_args([])


class _Groups:
    """The growing list of stamp groups plus a cursor; a camera's stamps are merged in one forward sweep."""

    def __init__(self, groups, thresh):
        self.groups, self.thresh, self.at = groups, thresh, 0

    def open(self):
        return self.at < len(self.groups)

    def offset(self, stamp):
        """stamp minus the mean of the group under the cursor"""
        g = self.groups[self.at]
        total = 0   # my_mean (Tools/timeAlign.py:12-16): naive left-to-right accumulation.  The builtin sum() of floats is
        for x in g:  # compensated (Neumaier) from Python 3.12 on and can differ in the last ulp, which flips the
            total += x   # abs(diff) < thresh comparison on the boundary
        return stamp - total / len(g)

    def take(self, stamp):
        self.groups[self.at].append(stamp)
        self.at += 1

    def merge(self, stamps):
        self.at = 0
        for stamp in stamps:
            if not self.open():
                return
            off = self.offset(stamp)
            if abs(off) < self.thresh:
                self.take(stamp)
            elif off > 0:
                # the stamp is newer than this group: walk on until a group is in reach, or already newer than the stamp
                while self.open() and off >= self.thresh:
                    self.at += 1
                    if self.open():
                        off = self.offset(stamp)
                if self.open() and abs(off) < self.thresh:
                    self.take(stamp)
            # off <= -thresh: older than every group still open -> the stamp has no partner and is dropped


def align_time(time_dict, thresh, init=True, info_list=None):
    """Greedy alignment of ascending time stamps (Tools/timeAlign.py:18-72).

    ``time_dict``: {camera: [t0, t1, ...]}.  ``init=True``: the camera whose FIRST stamp is the latest seeds one group per
    stamp.  ``init=False``: ``info_list = [groups, cams]`` from an earlier call is extended instead.  Returns
    ``(groups, cams)``; ``cams`` lists the cameras in the order their stamps sit inside a group."""
    if init is True:
        seed, newest = str(), 0
        for cam, stamps in time_dict.items():
            if stamps[0] > newest:
                seed, newest = cam, stamps[0]
        groups, cams = [[t] for t in time_dict[seed]], [seed]
    else:
        (groups, cams), seed = info_list, None
    sweep = _Groups(groups, thresh)
    for cam, stamps in time_dict.items():
        if cam != seed:
            cams.append(cam)
            sweep.merge(stamps)
    return groups, cams


class TimeParser(object):
    """Folders of ``<timestamp>.<ext>`` files -> the complete 4-camera groups (Tools/timeAlign.py:74-88)."""

    def __init__(self, args):
        self.cams = list(CAMERA_DIRS)
        self.usb_align_thresh = args.usb_align_thresh
        self.cam_dict = {cam: self.get_time_list(getattr(args, cam)) for cam in self.cams}

    def get_time_list(self, cam_dir):
        return sorted(float(name[:-4]) for name in os.listdir(cam_dir))   # "<stamp>.jpg" / "<stamp>.png"

    def usb_cam_align(self):
        groups, cams = align_time(self.cam_dict, self.usb_align_thresh, init=True, info_list=None)
        return [g for g in groups if len(g) == len(self.cams)], cams


def main():
    complete, order = TimeParser(args).usb_cam_align()
    print(len(complete))
    print(order)


if __name__ == '__main__':
    main()
