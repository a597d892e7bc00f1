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
import os
import pickle
import traceback

# blocks are named bnpc_<pid of the creating worker>_<random>: what a crashed
# run leaves in /dev/shm can be recognised (and swept) by the prefix
SHM_PREFIX = 'bnpc_'


def pack(obj):
    """(in-band pickle bytes, shared-memory block name, buffer sizes)"""
    buffers = []
    data = pickle.dumps(obj, protocol=5, buffer_callback=buffers.append)
    raws = [b.raw() for b in buffers]
    sizes = [r.nbytes for r in raws]
    block = None
    for _ in range(8):
        try:
            block = shared_memory.SharedMemory(create=True,
                size=max(1, sum(sizes)),
                name=f'{SHM_PREFIX}{os.getpid()}_{os.urandom(6).hex()}')
            break
        except FileExistsError:
            continue
    if block is None:
        block = shared_memory.SharedMemory(create=True,
            size=max(1, sum(sizes)))
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


def discard(packed):
    """Release the shared-memory block of a `pack` nobody will unpack."""
    try:
        block = shared_memory.SharedMemory(name=packed[1])
    except (FileNotFoundError, OSError):
        return
    block.close()
    try:
        block.unlink()
    except (FileNotFoundError, OSError):
        pass


def _worker(fn, job, conn):
    packed = None
    try:
        packed = pack(fn(*job))
        conn.send(('ok', packed))
        packed = None               # the parent owns the block now
    except BaseException as err:     # noqa: B902 - reported, then re-raised
        if packed is not None:      # the name never reached the parent
            discard(packed)
            packed = None
        try:
            conn.send(('error', f'{type(err).__name__}: {err}\n'
                + traceback.format_exc()))
        except (OSError, ValueError):   # the parent is gone
            pass
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
    try:
        while waiting:
            for conn in connection.wait(list(waiting)):
                proc = waiting.pop(conn)
                try:
                    kind, payload = conn.recv()
                except (EOFError, OSError):
                    proc.join()
                    failures.append(f'worker pid {proc.pid} ended with exit '
                        f'code {proc.exitcode} without a result')
                    continue
                finally:
                    conn.close()
                if kind == 'ok':
                    on_done(unpack(payload))    # (unpack unlinks the block)
                else:
                    failures.append(payload)
    finally:
        # on_done / unpack raised (or the caller was interrupted): the other
        # workers may have packed - or be about to pack - results nobody will
        # read.  Drain their pipes, release those blocks (GBs at config 4/5,
        # otherwise left in /dev/shm until reboot), leave no process behind.
        for conn, proc in waiting.items():
            try:
                if conn.poll(30):
                    kind, payload = conn.recv()
                    if kind == 'ok':
                        discard(payload)
            except (EOFError, OSError):
                pass
            finally:
                conn.close()
        for proc in procs:
            proc.join(30 if waiting else None)
            if proc.is_alive():
                proc.terminate()
                proc.join()
    return failures
