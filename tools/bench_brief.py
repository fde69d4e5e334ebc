#!/usr/bin/env python3
"""stdin: bench.py's output; prints a few fields of its JSON line (label = argv[1])."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else '', 'value %.1f serial %.1f' % (d['value'], d.get('value_serial') or 0), 'identical', d.get('replays_identical_to_eager'),
      'reproducible', d.get('step_bitwise_reproducible'), 'deterministic-library', d.get('library_deterministic_mode'))
