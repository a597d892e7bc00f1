import numpy as np, sys, os, subprocess
if len(sys.argv) > 1:
    from bnpc_amd import _lib
    rng=np.random.RandomState(0)
    N,M,K=[int(x) for x in sys.argv[1:4]]
    data=(rng.random_sample((N,M))<0.3).astype(float)
    ctx=_lib.Context(data=data)
    theta=np.clip(rng.uniform(size=(K,M)),1e-5,1-1e-5).astype(np.float32)
    out=ctx.ll_theta(0,theta,.01,.2)
    print(ctx.last_launch(), out[:1,:3],flush=True)
    sys.exit(0)
for env in ({'BNPC_FOLD_DEBUG':'4'},{'BNPC_FOLD_DEBUG':'0'}):
    for shape in ((700,130,14),(64,64,2)):
        e=dict(os.environ); e.update(env)
        r=subprocess.run([sys.executable,__file__]+[str(x) for x in shape],env=e,capture_output=True,text=True)
        print(env,shape,r.returncode,r.stdout.strip()[-200:],r.stderr.strip()[:150].replace('\n',' | '),flush=True)
