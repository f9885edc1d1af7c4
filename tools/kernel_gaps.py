#!/usr/bin/env python3
"""Gaps between the two launches of a single-FoV step in a rocprofv3 --kernel-trace
CSV: resident stack -> faces + paste + next conv0_a -> next stack."""
import csv
import statistics as st
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
  rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
rows.sort()
g1, g2, d1, d2 = [], [], [], []
for a, b in zip(rows, rows[1:]):
  if 'conv32ps' in a[2] and 'faces_paste_conv0a' in b[2]:
    g1.append(b[0] - a[1])
    d1.append(a[1] - a[0])
  if 'faces_paste_conv0a' in a[2] and 'conv32ps' in b[2]:
    g2.append(b[0] - a[1])
    d2.append(a[1] - a[0])
starts = [r[0] for r in rows if 'conv32ps' in r[2]]
per = [b - a for a, b in zip(starts, starts[1:]) if b - a < 400000]
q = lambda v, p: sorted(v)[int(p * (len(v) - 1))] / 1e3
print('stack %.1f us (median), fused launch %.2f us; gap stack -> fused %.2f us (90 %%: %.2f), '
      'fused -> next stack %.2f us (90 %%: %.2f); period %.1f us over %d steps' % (
          st.median(d1) / 1e3, st.median(d2) / 1e3, st.median(g1) / 1e3, q(g1, 0.9),
          st.median(g2) / 1e3, q(g2, 0.9), st.median(per) / 1e3, len(per)))
# the same as MEANS over the steps whose three launches follow each other (medians of different
# distributions do not add up to the period; means do)
seq = []
for a, b, c in zip(rows, rows[1:], rows[2:]):
  if 'conv32ps' in a[2] and 'faces_paste_conv0a' in b[2] and 'conv32ps' in c[2] and c[0] - a[0] < 400000:
    seq.append((a[1] - a[0], b[0] - a[1], b[1] - b[0], c[0] - b[1], c[0] - a[0]))
if seq:
  m = [sum(x[i] for x in seq) / len(seq) / 1e3 for i in range(5)]
  print('means over %d stack -> fused -> stack triples: stack %.2f us + gap %.2f + fused %.2f + gap %.2f = '
        'period %.2f us' % (len(seq), m[0], m[1], m[2], m[3], m[4]))
# a stack queued ahead whose conv0_a found no valid position ends after its first conv: those
# launches are in rocprofv3's per-kernel average, they are not stacks
d = [r[1] - r[0] for r in rows if 'conv32ps' in r[2]]
short = [x for x in d if x < 60000]
full = [x for x in d if x >= 60000]
if d:
  print('conv32ps launches: %d, of them %d shorter than 60 us (gave up after the first conv: mean %.1f us); '
        'mean of the others %.2f us (rocprofv3 --stats averages over all of them: %.2f us)' % (
            len(d), len(short), (sum(short) / len(short) / 1e3) if short else 0.0,
            sum(full) / max(len(full), 1) / 1e3, sum(d) / len(d) / 1e3))

