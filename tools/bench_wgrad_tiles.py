import os, sys, subprocess
if len(sys.argv) == 1:
    for t in (1, 2, 3):
        env = dict(os.environ, GPV_FORCE_WGRAD_TILE=str(t))
        print('== wgrad tile cfg', ['128x128', '128x64', '64x64'][t - 1]); sys.stdout.flush()
        subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'bench_split_conv.py')], env=env)
