"""rewrite the table between <!-- knobs:begin --> and <!-- knobs:end --> of INTEGRATION.md from tools/list_knobs.py"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'list_knobs.py')], capture_output=True, text=True, check=True).stdout.strip()
p = os.path.join(ROOT, 'INTEGRATION.md')
s = open(p).read()
new, n = re.subn(r'(<!-- knobs:begin -->\n).*?(\n<!-- knobs:end -->)', lambda m: m.group(1) + table + m.group(2), s, flags=re.S)
assert n == 1, 'markers not found'
open(p, 'w').write(new)
print('INTEGRATION.md: %d knob rows' % (table.count('\n') - 1))
