"""Chains to and from their worker processes.

The reference runs its chains in a ``multiprocessing.Pool`` and moves the model
in and the finished chain out by pickling them through the pool's pipes
(/root/reference/libs/MCMC.py:113-120, 159-161) - gigabytes at the large
configurations.  Here

  * every chain gets its own PROCESS, started by fork: the model (and, for
    lugsail extension rounds, the chain so far) is simply there, copy-on-write,
    no pickle in that direction;
  * a finished chain returns through POSIX shared memory: the worker pickles
    it with protocol 5, whose out-of-band buffers - the sample arrays - are
    written once into one shared-memory block; only the small in-band part and
    the block's name travel through the pipe; the parent maps the block,
    rebuilds the arrays and unlinks it;
  * a worker that raises reports the traceback; one that dies without a word
    (a GPU fault, the OOM killer) is noticed by its exit code.  Nothing is
    restarted: a process that has touched the GPU is not re-used.

If the parent has already initialised the HIP runtime (a model used in-process
before the run), forking would hand the children a runtime they cannot use, so
the workers are spawned instead (the model then travels by pickle, once).
"""
import multiprocessing as mp
from multiprocessing import connection, resource_tracker, shared_memory
import pickle
import traceback


def pack(obj):
    """(in-band pickle bytes, shared-memory block name, buffer sizes)"""
    buffers = []
    data = pickle.dumps(obj, protocol=5, buffer_callback=buffers.append)
    raws = [b.raw() for b in buffers]
    sizes = [r.nbytes for r in raws]
    block = shared_memory.SharedMemory(create=True, size=max(1, sum(sizes)))
    at = 0
    for raw, size in zip(raws, sizes):
        block.buf[at:at + size] = raw
        at += size
    name = block.name
    block.close()
    # the block now belongs to whoever unpacks it: without this the creating
    # process's resource tracker unlinks it again at exit ("leaked
    # shared_memory objects") - or, if it exits first, before it was read
    try:
        resource_tracker.unregister(block._name, 'shared_memory')
    except Exception:       # noqa: BLE001 - bookkeeping only
        pass
    return data, name, sizes


def unpack(packed):
    """The object of `pack`; the shared-memory block is released."""
    data, name, sizes = packed
    block = shared_memory.SharedMemory(name=name)
    try:
        views, at = [], 0
        for size in sizes:
            # private, writable copies: the block is unlinked right away
            views.append(bytearray(block.buf[at:at + size]))
            at += size
        return pickle.loads(data, buffers=views)
    finally:
        block.close()
        block.unlink()


def _worker(fn, job, conn):
    try:
        conn.send(('ok', pack(fn(*job))))
    except BaseException as err:     # noqa: B902 - reported, then re-raised
        conn.send(('error', f'{type(err).__name__}: {err}\n'
            + traceback.format_exc()))
        raise
    finally:
        conn.close()


def run_jobs(fn, jobs, on_done):
    """fn(*job) in one process per job; on_done(result) in this process as
    results arrive.  Returns the list of failure descriptions (empty = all
    workers finished)."""
    from bnpc_amd import _lib
    ctx = mp.get_context('spawn' if _lib.gpu_touched() else 'fork')
    waiting = {}
    for job in jobs:
        recv_end, send_end = ctx.Pipe(duplex=False)
        proc = ctx.Process(target=_worker, args=(fn, job, send_end))
        proc.start()
        send_end.close()
        waiting[recv_end] = proc
    failures = []
    procs = list(waiting.values())
    while waiting:
        for conn in connection.wait(list(waiting)):
            proc = waiting.pop(conn)
            try:
                kind, payload = conn.recv()
            except (EOFError, OSError):
                proc.join()
                failures.append(f'worker pid {proc.pid} ended with exit code '
                    f'{proc.exitcode} without a result')
                continue
            finally:
                conn.close()
            if kind == 'ok':
                on_done(unpack(payload))
            else:
                failures.append(payload)
    for proc in procs:
        proc.join()
    return failures
