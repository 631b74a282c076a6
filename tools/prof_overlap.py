"""One forward of an overlapped (multi-stream) run from a rocprofv3 rocpd database: every kernel with start / end relative to the
step, the stream it ran on, and the union of busy time (how much of the step had at least one kernel running).

    python tools/prof_overlap.py /tmp/prof/p_results.db [--step-kernel multi_copy_kernel]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--step-kernel', default='multi_copy_kernel', help='a kernel launched exactly once per forward (marks the step)')
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    rows = c.execute('select name, start, end%s from kernels order by start' % ((', ' + qcol) if qcol else '')).fetchall()
    short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    marks = [i for i, r in enumerate(rows) if short(r[0]).startswith(a.step_kernel)]
    i0, i1 = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
    seq = rows[i0:i1]
    t0 = seq[0][1]
    busy, cur_end = 0, 0
    for r in seq:
        s, e = r[1], r[2]
        if s > cur_end:
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
        print('%8.1f %8.1f  %7.1f us  q%-4s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r[3] if qcol else '-', short(r[0])[:70]))
    wall = max(r[2] for r in seq) - t0
    print('step: %.1f us wall, %.1f us with at least one kernel running, %.1f us sum of kernel durations' % (wall / 1e3, busy / 1e3, sum(r[2] - r[1] for r in seq) / 1e3))


if __name__ == '__main__':
    main()
