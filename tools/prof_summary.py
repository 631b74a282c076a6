"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel stats as CSV, and the timeline of one forward
(the last complete graph replay: kernels between two consecutive launches of the first kernel of the forward).

    python tools/prof_summary.py gpurun_out/prof/x_results.db [--csv profiles/r01_kernel_stats.csv] [--timeline]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--csv')
    ap.add_argument('--timeline', action='store_true')
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')
    stats = {}
    for n, s, e in rows:
        d = stats.setdefault(short(n), [])
        d.append(e - s)
    tot = sum(sum(v) for v in stats.values())
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for n, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
    if a.csv:
        open(a.csv, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(l[:200] for l in lines[:40]))
    if a.timeline:
        first = short(rows[0][0])
        # a forward = from one launch of the kernel that starts it to the next one
        starts = [i for i, r in enumerate(rows) if short(r[0]).startswith('pad_heads_kernel')]
        # two pad_heads launches (weight, bias) per encoder per forward: take a window of one forward in the middle
        names = [short(r[0]) for r in rows]
        # find period: index distance between repeats of the whole sequence
        period = None
        for p in range(20, 1500):
            mid = len(rows) // 2
            if names[mid:mid + p] == names[mid + p:mid + 2 * p]:
                period = p
                break
        print('period (launches per forward): %s' % period)
        if period:
            mid = len(rows) // 2
            # align to the launch after the largest gap within a period (the step boundary)
            win = rows[mid:mid + 2 * period]
            gaps = [(win[i + 1][1] - win[i][2], i) for i in range(period)]
            g, i0 = max(gaps)
            seq = win[i0 + 1:i0 + 1 + period]
            t0 = seq[0][1]
            busy = 0
            for n, s, e in seq:
                print('%9.1f us  +%8.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, short(n)[:110]))
                busy += e - s
            print('forward: %.1f us wall, %.1f us kernel-busy (sum of durations)' % ((seq[-1][2] - t0) / 1e3, busy / 1e3))


if __name__ == '__main__':
    main()
