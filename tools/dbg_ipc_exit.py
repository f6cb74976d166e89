"""Two processes, IPC-window communicator, explicit teardown orders (debug helper for the exit-time heap message)."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch.multiprocessing as mp


def worker(rank, port, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0')
    import torch, torch.distributed as dist
    from pauxy_amd.comm import TorchComm
    from pauxy_amd import systems, trial as trial_mod, _lib as L
    from pauxy_amd.qmc.afqmc import AFQMC
    dist.init_process_group('gloo', rank=rank, world_size=2)
    comm = TorchComm(device=torch.device('cpu'))
    s = systems.synthetic_generic(12, 10, (3, 3), seed=3)
    t = trial_mod.rhf_trial_generic(s)
    o = {'qmc': {'timestep': 0.01, 'num_steps': 10, 'blocks': 2, 'stabilise_freq': 5, 'pop_control_freq': 5, 'num_walkers': 12},
         'propagator': {'device_rng': True}, 'estimators': {'mixed': {'energy_eval_freq': 2, 'verbose': False}},
         'walkers': {'device_comm': 'ipc'}}
    afqmc = AFQMC(comm=comm, options=o, system=s, trial=t)
    afqmc.run_batched()
    st = afqmc.psi.dev.comm_stats()
    sys.stderr.write("rank %d mode %s stats %r\n" % (rank, mode, st))
    dist.barrier()
    if mode == 'destroy':
        afqmc.psi.dev.comm_destroy()
        dist.barrier()
        afqmc.psi.dev.close()
    elif mode == 'close':
        afqmc.psi.dev.close()
        dist.barrier()
    q.put(rank)
    dist.barrier()
    dist.destroy_process_group()
    sys.stderr.write("rank %d mode %s leaving\n" % (rank, mode))


if __name__ == '__main__':
    for mode in ('destroy', 'close', 'none'):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, port, mode, q)) for r in range(2)]
        for p in ps: p.start()
        for p in ps: p.join(120)
        print("mode", mode, "exit codes", [p.exitcode for p in ps], flush=True)
