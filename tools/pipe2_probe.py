"""dev tool: which co-scheduling of consecutive pairs gives the best throughput?  The bench's default is three whole forwards
round-robin on three streams; here the forward is cut into stages that run on dedicated streams (a software pipeline across
pairs), so that e.g. the trunk of pair k+1 (short, under-filled kernels) always runs beside the FPN / RPN / heads of pair k."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereo_rcnn_amd import fixture, engine, _lib
from stereo_rcnn_amd.model.stereo_rcnn.resnet import resnet

dev = torch.device('cuda:0')
m = resnet(('__background__', 'Car'), 101); m.create_architecture()
m.load_state_dict(fixture.make_state_dict(3)); m.cuda(); m.eval(); m.precision = 'f16x3'
l, r, info = [t.to(dev) for t in fixture.make_inputs(3, 375, 1242)]
engine.PRECISION = 'f16x3'
NSLOT = 4
plans = [m._get_plan(1, 600, 1987, s) for s in range(NSLOT)]
with torch.no_grad():
    for pl in plans:
        pl.fmt = _lib.FMT_SPLIT16
        pl.set_inputs(l, r, info)
        pl.launch_all()
        torch.cuda.synchronize()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 90

    def whole(ns):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        def run(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % ns]):
                    pl = plans[k % ns]
                    pl.set_inputs(l, r, info)
                    pl.launch_all()
        return run

    def staged(cuts, nslot, prio=None):
        """cuts: list of lists of stage names; one stream per list"""
        streams = [torch.cuda.Stream(priority=(prio[i] if prio else 0)) for i in range(len(cuts))]
        done = {}
        def run(n):
            for k in range(n):
                pl = plans[k % nslot]
                prev = None
                for si, (s, names) in enumerate(zip(streams, cuts)):
                    with torch.cuda.stream(s):
                        if si == 0 and k >= nslot:
                            s.wait_event(done[k - nslot])          # the slot's buffers are free again
                        if prev is not None:
                            s.wait_event(prev)
                        if si == 0:
                            pl.set_inputs(l, r, info)
                        for nm in names:
                            getattr(pl, nm)()
                        prev = torch.cuda.Event(); prev.record(s)
                done[k] = prev
                done.pop(k - 2 * nslot, None)
        return run

    modes = [('whole forwards, 1 stream', whole(1)), ('whole forwards, 2 streams', whole(2)), ('whole forwards, 3 streams', whole(3)),
             ('trunk | fpn_rpn+proposals+heads, 2 slots', staged([['trunk'], ['fpn_rpn', 'proposals', 'heads']], 2)),
             ('trunk | fpn_rpn+proposals+heads, 3 slots', staged([['trunk'], ['fpn_rpn', 'proposals', 'heads']], 3)),
             ('trunk | fpn_rpn+proposals | heads, 3 slots', staged([['trunk'], ['fpn_rpn', 'proposals'], ['heads']], 3)),
             ('trunk | fpn_rpn+proposals | heads, 4 slots', staged([['trunk'], ['fpn_rpn', 'proposals'], ['heads']], 4)),
             ('trunk(high prio) | rest, 3 slots', staged([['trunk'], ['fpn_rpn', 'proposals', 'heads']], 3, prio=[-1, 0])),
             ('trunk | rest(high prio), 3 slots', staged([['trunk'], ['fpn_rpn', 'proposals', 'heads']], 3, prio=[0, -1]))]
    for name, run in modes:
        run(6); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(N); th = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('%-50s %.3f ms/pair = %.1f pairs/s   (host enqueue %.2f ms/pair, eager Python)' % (name, dt * 1e3 / N, N / dt, th * 1e3 / N), flush=True)
