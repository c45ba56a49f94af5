"""PNG decode worker PROCESS of the KITTI loop (test_net.run_split): run as a script, never imported by the package.

    python png_worker.py <shared file> <slot> <entry bytes>

Reads one JSON request per line on stdin ({"paths": [left.png, right.png]}), decodes each file exactly as test_net.read_png_rgb does
(PIL, RGB, uint8) into its slot of the shared file (slot * 2 entries of `entry bytes`), and answers one JSON line ({"shapes": [[H, W, 3],
[H, W, 3]]} or {"error": ...}).  Sixteen decoder THREADS inside the loop's process cost the streamed flow 1.1 ms per pair -- not their CPU
time (21 ms per pair, spread over 16 threads) but their hundreds of GIL hand-overs per image against the loop thread's launches
(tools/config3_ablate.py: 138 pairs/s as shipped, 164 with the decode taken out); a process has its own interpreter.  Imports numpy and
PIL only (no torch, no HIP): starts in ~0.2 s."""
import json
import mmap
import sys


def main(argv):
    import numpy as np
    from PIL import Image
    path, slot, nbytes = argv[1], int(argv[2]), int(argv[3])
    with open(path, 'r+b') as fh:
        mm = mmap.mmap(fh.fileno(), 0)
    base = slot * 2 * nbytes
    sys.stdout.write('ready\n')
    sys.stdout.flush()
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            shapes = []
            for k, p in enumerate(json.loads(line)['paths']):
                with Image.open(p) as im:
                    a = np.asarray(im.convert('RGB'), dtype=np.uint8)
                if a.nbytes > nbytes:
                    raise ValueError('image of %d bytes does not fit the %d-byte entry' % (a.nbytes, nbytes))
                np.frombuffer(mm, dtype=np.uint8, count=a.nbytes, offset=base + k * nbytes)[:] = a.reshape(-1)
                shapes.append(list(a.shape))
            reply = {'shapes': shapes}
        except Exception as e:           # reported to the parent, which raises
            reply = {'error': '%s: %s' % (type(e).__name__, e)}
        sys.stdout.write(json.dumps(reply) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main(sys.argv)
