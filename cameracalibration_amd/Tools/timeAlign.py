"""Timestamp alignment of the four camera directories -- the feeder that groups front/back/left/right files into
4-camera frame sets for BevGenerator.batch().  Same behaviour and names as the reference's Tools/timeAlign.py
(align_time :18-72, TimeParser :74-88); host-side list logic, no pixels involved.
"""
from __future__ import annotations

import argparse
import os

parser = argparse.ArgumentParser(description="Time Align for Images")
parser.add_argument("--front", type=str, default="./data/front")
parser.add_argument("--back", type=str, default="./data/back")
parser.add_argument("--left", type=str, default="./data/left")
parser.add_argument("--right", type=str, default="./data/right")
parser.add_argument("--usb_align_thresh", type=float, default=0.1)
args = parser.parse_args([])


def _centre(group):
    return sum(group) / len(group)


def align_time(time_dict, thresh, init=True, info_list=None):
    """Greedy alignment (Tools/timeAlign.py:18-72).

    time_dict: {camera: ascending timestamps}.  With init=True the camera whose FIRST stamp is the latest seeds one
    group per stamp; otherwise info_list = [groups, cams] is extended.  Every other camera walks its stamps and the groups
    in lock-step: a stamp within `thresh` of the current group's mean joins it; a stamp that is later skips groups
    until one is within reach (or later than the stamp); a stamp that is earlier than the current group is dropped.
    Returns (groups, cams) with cams in the order the cameras were merged."""
    if init is True:
        seed = None
        latest = 0
        for cam, stamps in time_dict.items():
            if stamps[0] > latest:
                latest, seed = stamps[0], cam
        if seed is None:
            seed = str()
        groups = [[t] for t in time_dict[seed]] if seed in time_dict else []
        cams = [seed]
    else:
        groups, cams = info_list
        seed = None
    for cam, stamps in time_dict.items():
        if cam == seed:
            continue
        cams.append(cam)
        g = 0
        for t in stamps:
            if g >= len(groups):
                break
            delta = t - _centre(groups[g])
            if abs(delta) < thresh:
                groups[g].append(t)
                g += 1
                continue
            if delta <= 0:
                continue  # this stamp is older than every group still open: nothing to pair it with
            reached = True
            while not (delta < 0 or abs(delta) < thresh):
                g += 1
                if g >= len(groups):
                    reached = False
                    break
                delta = t - _centre(groups[g])
            if reached and abs(delta) < thresh:
                groups[g].append(t)
                g += 1
    return groups, cams


class TimeParser(object):
    """Tools/timeAlign.py:74-88: directories of <timestamp>.<ext> files -> complete 4-camera groups."""

    def __init__(self, args):
        self.cams = ["front", "back", "left", "right"]
        self.usb_align_thresh = args.usb_align_thresh
        self.cam_dict = {cam: self.get_time_list(getattr(args, cam)) for cam in self.cams}

    def get_time_list(self, cam_dir):
        return sorted(float(name[:-4]) for name in os.listdir(cam_dir))

    def usb_cam_align(self):
        groups, cams = align_time(self.cam_dict, self.usb_align_thresh, init=True, info_list=None)
        return [g for g in groups if len(g) == 4], cams


def main():
    res, base = TimeParser(args).usb_cam_align()
    print(len(res))
    print(base)


if __name__ == '__main__':
    main()
