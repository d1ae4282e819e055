"""CPU restatement of the reference's event loop, main.py:138-168 - TEST INFRASTRUCTURE, not product code.

Only tests/ may import this file.  It restates the control flow of the loop as a pure function: which sensor record is
consumed at which merged timestamp, with which dt, which scans pass the motion gate, which of them are matched against
the previous scan (`adj`), when `last_scan` is refreshed.  What the particles do is not its business: the pose of
particle 0 in front of every lidar event comes in as data (recorded from the run under test).

Faithful to the reference, quirks included:
  * times = np.unique(concatenate(imu times, lidar times)) (main.py:114); a cursor advances only when the timestamp of
    the record under it EQUALS the current merged time (:139-140,147-149) and is clamped to the last record: a
    duplicated or decreasing timestamp in a sensor's array stalls that sensor for the rest of the run (data/intel.txt
    through the Intel adapters does that; the Freid101-family adapters hand over unique, sorted times);
  * dt of an IMU event = its timestamp minus the previous IMU event's, the first is 0 (:102,141-145);
  * the gate (:152-155): update_count < 2, or particle 0 moved >= 0.33 m or turned >= pi/9 (plain difference of the
    headings, no wrap) since the last pose that passed the gate BY MOTION;
  * frame numbering: plotFrameNumber counts lidar events, accepted or not, starting at 1550 (:110) - a multiple of 5,
    so `frame % 5 < 2` (:156) and `frame % 5 == 0` (:167) behave as if it started at 0;
  * the cold start: the committed script restores pickle/1550.state and starts at t_idx = 3422 (:115-136); the
    restatement starts at the first merged time with the initial values of :102-112.
"""
from math import pi, sqrt

import numpy as np

MAX_UPDATE_COUNT = 2          # main.py:41
ROT_THRESHOLD = pi / 9        # main.py:42
DIST_THRESHOLD = 0.33         # main.py:43


def replay_decisions(imu_times, lidar_times, pose0_before_scan, first_pose=(0.0, 0.0, 0.0), start_frame=1550, max_frames=None):
    """Returns the list of events the reference's loop would process:
      ("imu", imu index, dt)
      ("scan", lidar index, frame number, accepted, adj, refresh_last_scan)
    `pose0_before_scan(k)` -> (x, y, theta) of particle 0 when the k-th lidar event (k = 0, 1, ...) is reached."""
    imu_times = np.asarray(imu_times)
    lidar_times = np.asarray(lidar_times)
    times = np.unique(np.concatenate((imu_times, lidar_times)))                # main.py:114
    prev_timestamp = imu_times[0]                                               # :102
    imu_idx = lidar_idx = 0                                                     # :103-104
    frame = start_frame                                                         # :110
    last_updated_pose = tuple(first_pose)                                       # :111
    update_count = 0                                                            # :113
    events = []
    n_scan_events = 0
    for t in times:                                                             # :138
        if imu_times[imu_idx] == t:                                             # :139-140
            ts = imu_times[imu_idx]
            this = imu_idx
            imu_idx = min(imu_idx + 1, len(imu_times) - 1)                      # :141
            events.append(("imu", int(this), int(ts - prev_timestamp)))         # :142-144
            prev_timestamp = ts                                                 # :145
        if lidar_times[lidar_idx] == t:                                         # :147
            this = lidar_idx
            lidar_idx = min(lidar_idx + 1, len(lidar_times) - 1)                # :149
            curr = pose0_before_scan(n_scan_events)                             # :152
            dist = sqrt((last_updated_pose[0] - curr[0]) ** 2 + (last_updated_pose[1] - curr[1]) ** 2)   # :153
            rot = abs(last_updated_pose[2] - curr[2])                           # :154
            accepted = update_count < MAX_UPDATE_COUNT or dist >= DIST_THRESHOLD or rot >= ROT_THRESHOLD   # :155
            adj = refresh = False
            if accepted:
                adj = not (frame % 5 < 2)                                       # :156-159
                if dist >= DIST_THRESHOLD or rot >= ROT_THRESHOLD:              # :162-164
                    update_count = 0
                    last_updated_pose = tuple(curr)
                elif update_count < MAX_UPDATE_COUNT:                           # :165-166
                    update_count += 1
                refresh = frame % 5 == 0                                        # :167
            events.append(("scan", int(this), int(frame), bool(accepted), bool(adj), bool(refresh)))
            frame += 1                                                          # :183
            n_scan_events += 1
            if max_frames is not None and n_scan_events >= max_frames:
                break
    return events


def occupied_points(tiles, cell_size, threshold=1.0):
    """HybridMap.get_occupied_points (hybridmap.py:303-313): x and y of every cell with log-odds > threshold, in CELL units, tile after tile.  `tiles` = iterable of
    ((cx, cy), float map[dim][dim]) as the oracle maps hold them."""
    xs, ys = [], []
    for (cx, cy), m in tiles:
        dim = m.shape[0]
        i, j = np.nonzero(m > threshold)
        xs.append(((i - dim / 2) * cell_size + cx) / cell_size)
        ys.append(((j - dim / 2) * cell_size + cy) / cell_size)
    return (np.concatenate(xs) if xs else np.empty(0)), (np.concatenate(ys) if ys else np.empty(0))
