import sys, time, numpy as np, torch, os
os.environ['TQDM_DISABLE']='1'
sys.path.insert(0,'.')
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table, cjk_vocab
from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
for V in (3000, 20000, 50000):
    X = lorentz_table(V, 100, seed=42, scale=0.05)
    table = torch.zeros((V+64, 101), device='cuda'); table[:V]=X.cuda()
    eng = MergeEngine(V+64, 101, 'reference'); eng.set_table(table, V)
    t0=time.time(); a=eng.argmin(1.0, 1e-5); t1=time.time(); print(V,'argmin',a, eng.scan_stats(), round(1e3*(t1-t0),2),'ms')
    t0=time.time(); d,i,j,c=eng.topk(1.0,1e-5,10000); t1=time.time()
    print(V,'topk n',len(d),'count',c,'expected',V*(V-1)//2, eng.scan_stats(), round(1e3*(t1-t0),2),'ms')
    # expected: first 10000 pairs in row-major order, all d == 0
    exp=[]; ii=0
    while len(exp)<10000:
        for jj in range(ii+1,V):
            exp.append((ii,jj))
            if len(exp)==10000: break
        ii+=1
    ok = list(zip(i.tolist(),j.tolist()))==exp and (d==0).all()
    print('   order ok', ok)
    t0=time.time(); d2,i2,j2,c2=eng.topk(1.0,1e-5,10000); t1=time.time(); print('   second', c2==c, np.array_equal(i2,i), eng.scan_stats(), round(1e3*(t1-t0),2),'ms')
# literal-mode fast tokenizer at 20k
V=20000
X = lorentz_table(V, 50, seed=42, scale=0.05)
import random; random.seed(1)
tok=FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), merge_threshold=0.1, sign_convention='reference', max_vocab_size=V+400)
t0=time.time(); tok.optimize_merges(steps=250, log_every=1000); print('literal fast 250 steps', round(time.time()-t0,2),'s', tok.merge_threshold)
h=[(tok.token2idx[a] if False else a,b) for a,b,_ in tok.merge_history[:6]]
print([ (tok.vocab.index(a), tok.vocab.index(b)) for a,b,_ in tok.merge_history[:5]], len(tok.merge_history))
