"""A few stages, one host thread each, items handed from stage to stage through short queues.

What the two jobs use to overlap their host stages with the GPU (the reference gets the same
effect from Spark running many tasks at once, /root/reference/src/jobs/prophet_modeler.py:139-141):

    modeler   read + parse the files of chunk k + 1  |  pack, fit (GPU), blobs of chunk k  |  parquet part k - 1
    scorer    read the models of chunk k + 1         |  predict (GPU) chunk k              |  CSV part k - 1

Every heavy step is a native call that releases the GIL (ctypes, pyarrow), so the threads really run side by side.
Items stay in order; the first exception stops the run and is re-raised in the caller.
"""
import os
import queue
import sys
import threading
import time

_STOP = object()


def clear_directory(path):
    """mode='overwrite' of the reference's writers (prophet_modeler.py:123-125, prophet_scorer.py:148-150): `path` becomes an
    empty directory.  What was there is moved aside at once and deleted on a thread of its own -- unlinking the previous
    run's part files (tens of megabytes of page cache to give back) took 7-10 ms of a 10 000-series job --; the returned
    function waits for that thread (call it before the job returns: nothing of the old run is left behind)."""
    import shutil
    old = None
    if os.path.isdir(path) or os.path.lexists(path):
        old = '%s.old-%d-%d' % (path.rstrip('/'), os.getpid(), threading.get_ident())
        try:
            os.rename(path, old)
        except OSError:
            old = None
            shutil.rmtree(path, ignore_errors=True)
    os.makedirs(path, exist_ok=True)
    if old is None:
        return lambda: None
    t = threading.Thread(target=lambda: shutil.rmtree(old, ignore_errors=True) if os.path.isdir(old) else os.unlink(old),
                         daemon=True)
    t.start()
    return t.join


class Laps(object):
    """TSF_JOB_TIMING=1 (dev): wall-clock laps of a host stage on stderr -- `with Laps('fit 2500') as lap: ...; lap('pack')`."""
    on = bool(os.environ.get('TSF_JOB_TIMING'))

    def __init__(self, what):
        self.what, self.t, self.out = what, time.time(), []

    def __call__(self, name):
        if self.on:
            now = time.time()
            self.out.append('%s %.1f' % (name, (now - self.t) * 1e3))
            self.t = now

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.on and self.out:
            sys.stderr.write('[job-timing] %s: %s ms\n' % (self.what, ', '.join(self.out)))
        return False


def run_pipeline(source, stages, depth=2):
    """source: an iterable of items, consumed on a thread of its own (a generator that does work -- reading a chunk --
    is thereby the first stage).  stages: callables item -> item, one thread each.
    depth: items that may wait between two stages.  Returns the list of the last stage's results, in order."""
    n = len(stages)
    qs = [queue.Queue(maxsize=depth) for _ in range(n + 1)]
    err = []
    stop = threading.Event()
    # TSF_PIPELINE_TIMING=1 (dev): when every stage had every item, in ms since the start, on stderr
    timing = [] if os.environ.get('TSF_PIPELINE_TIMING') else None
    t_0 = time.time()

    def note(stage, k, t_a):
        if timing is not None:
            timing.append((stage, k, (t_a - t_0) * 1e3, (time.time() - t_0) * 1e3))

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def feed():
        try:
            it, k = iter(source), 0
            while True:
                t_a = time.time()
                try:
                    item = next(it)
                except StopIteration:
                    break
                note(0, k, t_a)
                k += 1
                if not put(qs[0], item):
                    return
        except BaseException as e:        # noqa: BLE001 -- handed to the caller below
            err.append(e)
            stop.set()
        finally:
            put(qs[0], _STOP)

    def work(i):
        try:
            k = 0
            while True:
                try:
                    item = qs[i].get(timeout=0.05)
                except queue.Empty:
                    if stop.is_set():
                        return
                    continue
                if item is _STOP:
                    break
                t_a = time.time()
                res = stages[i](item)
                note(i + 1, k, t_a)
                k += 1
                if not put(qs[i + 1], res):
                    return
        except BaseException as e:        # noqa: BLE001
            err.append(e)
            stop.set()
        finally:
            put(qs[i + 1], _STOP)

    threads = [threading.Thread(target=feed, daemon=True)] + \
              [threading.Thread(target=work, args=(i,), daemon=True) for i in range(n)]
    # A stage that wakes up with an item has to wait for the interpreter lock until whoever runs Python code lets go of it
    # -- by default only after 5 ms, as long as a whole stage takes here.  Half a millisecond while the stages run.
    switch = sys.getswitchinterval()
    sys.setswitchinterval(min(switch, 0.0005))
    try:
        for t in threads:
            t.start()
        out = []
        while True:
            try:
                item = qs[n].get(timeout=0.05)
            except queue.Empty:
                if stop.is_set():
                    break
                continue
            if item is _STOP:
                break
            out.append(item)
        if err:
            stop.set()
        for t in threads:
            t.join()
    finally:
        sys.setswitchinterval(switch)
    if err:
        raise err[0]
    if timing is not None:
        for stage, k, a, b in sorted(timing):
            sys.stderr.write('[pipeline] stage %d item %d: %7.1f .. %7.1f ms (%.1f)\n' % (stage, k, a, b, b - a))
        sys.stderr.write('[pipeline] done at %.1f ms\n' % ((time.time() - t_0) * 1e3))
    return out
