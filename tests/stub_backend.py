"""TEST INFRASTRUCTURE: a device that does nothing, with edge264_amd.backend's Python surface, so that bench.py's rank
launcher, stream sharding and max-over-ranks reduction can run as N gloo ranks on a machine without GPUs
(E264_BENCH_BACKEND=tests.stub_backend).  It never produces samples: bench.py skips verification and baselines for it."""
import time

IS_STUB = True
RUN_RECON, RUN_DEBLOCK, RUN_ALL = 1, 2, 3
MAX_LANES = 4


class _Packet:
    def __init__(self, n):
        self.nbytes = n

    def free(self):
        pass


class Device:
    def __init__(self, ordinal=0):
        self.ordinal, self.launches, self.timing = ordinal, 0, False

    def close(self):
        pass

    def sync(self):
        pass

    def set_option(self, name, value):
        return 0

    def upload_packet(self, pkt):
        return _Packet(len(pkt))

    def make_batch(self, streams, packets):
        return (len(streams),)

    def submit_prepared(self, batch, mode=RUN_ALL):
        time.sleep(0.0005)
        if self.timing:
            self.launches += 1

    def free_batch(self, batch):
        pass

    def event_record(self, idx):
        pass

    def event_elapsed_ms(self, a, b):
        return 1.0

    def event_done(self, idx):
        return True

    def kernel_timing(self, enable):
        self.timing = bool(enable)
        if enable:
            self.launches = 0

    def kernel_time_ms(self):
        return [0.1 * self.launches, 0.4 * self.launches, 0.2 * self.launches, 0.3 * self.launches], self.launches


class Stream:
    def __init__(self, dev, w, h):
        pass

    def close(self):
        pass

    def bind_lane(self, lane):
        pass

    def alloc(self, slot, mirror=False):
        pass

    def fill(self, slot, value):
        pass
