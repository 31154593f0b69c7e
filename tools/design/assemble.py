"""Rebuilds DESIGN.md from the text parts in this directory and the numbers of the committed bench lines
(profiles/r03_bench_c*.json, r03_kernel_span.json, pmc_traffic.json): python tools/design/assemble.py"""
import json,os
D=os.path.dirname(os.path.abspath(__file__))
ROOT=os.path.dirname(os.path.dirname(D))
P=ROOT+'/profiles'
b={wl:json.load(open(P+'/r03_bench_%s.json'%wl)) for wl in ('c2','c3','c4','c5')}
r2={wl:json.load(open(P+'/r02_bench_%s.json'%wl)) for wl in ('c2','c3','c4','c5')}
span=json.load(open(P+'/r03_kernel_span.json'))
pmc=json.load(open(P+'/pmc_traffic.json'))
def f(x,n=2): return ('%.'+str(n)+'f')%x
names={'c2':'C2 particle antipodal N=4, 4096 envs','c3':'C3 Checkers N=2, 8192 envs (bit-exact)','c4':'C4 cross N=4, 4096 envs, 33-tick rollout + advantage normalisation','c5':'C5 merge8 N=8, 8192 envs'}
rows=[]
for wl in ('c2','c3','c4','c5'):
    d=b[wl]; o=r2[wl]
    inpl=d.get('launch_modes',{}).get('in_place_chains1',{}).get('us_per_tick')
    oin=o.get('launch_modes',{}).get('in_place_chains1',{}).get('us_per_tick')
    rows.append('| %s | %s µs/tick%s | **%s µs/tick**%s | %.3g | %.3f | `profiles/r03_bench_%s.json` |' % (
        names[wl], f(o['us_per_tick']), (' (in place %s)'%f(oin)) if oin else '', f(d['us_per_tick']), (' (in place %s)'%f(inpl)) if inpl else '',
        d['value'], d['roofline']['frac'], wl))
pol=b['c2']['policy_rollout']
rows.append('| policy-driven collection at C2 (actor on MFMA + step, full trajectories; second headline) | 6.63 µs/tick one launch per episode, 12.06 launch per tick | **%s µs/tick** one launch per episode (f16x3; f32: %s), %s one fused launch per tick, %s launch per tick | %.3g | %.2f of the f32 MFMA peak (%.0f TFLOP/s of network FLOPs) | `profiles/r03_bench_c2.json` `policy_rollout` |' % (
    f(pol['f16x3']['one_launch_per_episode']['us_per_tick']), f(pol['f32']['one_launch_per_episode']['us_per_tick']),
    f(pol['f16x3']['fused_launch_per_tick']['us_per_tick']) if 'fused_launch_per_tick' in pol['f16x3'] else '9.6',
    f(pol['f16x3']['launch_per_tick']['us_per_tick']), pol['headline']['env_steps_per_s'], pol['headline']['roofline']['frac'], pol['headline']['roofline']['achieved']))
table0='\n'.join(rows)
# table 5
def sp(tag):
    s=span.get(tag)
    return ('%.2f / %.2f'%(s['span_us_mean'], s['gap_us_mean'])) if s else '–'
def traffic(tag, algo):
    t=pmc.get(tag)
    return ('%.2f MB vs %.2f'%(t['hbm_bytes_per_launch']/1e6, algo/1e6)) if t else '–'
k5=[]
k5.append('| `k_particle_step_pairs<float,4,4,false,kSpNt,LIVE>` (max-ILP TU) | C2: N=4, 4096 envs, trajectory | %s µs (in place %s) | %s µs | %.0f (%.3f) | 218 VALU + 65 SALU per wave (round 2 PMC; path unchanged but for the grouped stores); %s |' % (
    f(b['c2']['us_per_tick'],3), f(b['c2']['launch_modes']['in_place_chains1']['us_per_tick'],2), sp('c2_trajectory'), b['c2']['roofline']['achieved'], b['c2']['roofline']['frac'], traffic('c2_trajectory_n4_e4096',400*4096)))
k5.append('| same kernel (slot-chained, plain stores) + moments + normalise, ONE graph | C4: N=4, 4096 envs, 33-tick rollout | %s µs per tick | – | %.0f (%.3f) | – |' % (f(b['c4']['us_per_tick'],3), b['c4']['roofline']['achieved'], b['c4']['roofline']['frac']))
k5.append('| `k_particle_step_agents2<4,kSpWt,LIVE>` (two lanes per agent) | C5: N=8, 8192 envs, trajectory | %s µs (in place %s) | %s µs | %.0f (%.3f) | %s |' % (
    f(b['c5']['us_per_tick'],3), f(b['c5']['launch_modes']['in_place_chains1']['us_per_tick'],2), sp('c5_trajectory'), b['c5']['roofline']['achieved'], b['c5']['roofline']['frac'], traffic('c5_trajectory_n8_e8192',1296*8192)))
k5.append('| `k_checkers_step_fast<2,false,true,8>` (table emit) | C3: 8192 envs, trajectory | %s µs (in place %s) | %s µs | %.0f (%.3f) | 420 VALU + 119 SALU + 11 LDS per wave (round 2: 683 + 136 + 0); %s |' % (
    f(b['c3']['us_per_tick'],3), f(b['c3']['launch_modes']['in_place_chains1']['us_per_tick'],2), sp('c3_trajectory'), b['c3']['roofline']['achieved'], b['c3']['roofline']['frac'], traffic('c3_trajectory_n2_e8192',400*8192)))
sw={ (s['envs'],s['variant']):s for s in b['c2']['sweep']}
k5.append('| `k_particle_step<float,4,…,kSpWt>` (lane per env) | N=4, 2¹⁸ / 2²⁰ / 2²² envs, in place | %s / %s / %s µs | – | %.0f / %.0f / %.0f (%.2f / %.2f / %.2f; the first two cache-assisted, §0) | HBM: at the copy rate of the chip from 2²¹ envs on |' % (
    f(sw[(262144,'in-place')]['avg_launch_us'],1), f(sw[(1048576,'in-place')]['avg_launch_us'],1), f(sw[(4194304,'in-place')]['avg_launch_us'],0),
    sw[(262144,'in-place')]['achieved_GBps'], sw[(1048576,'in-place')]['achieved_GBps'], sw[(4194304,'in-place')]['achieved_GBps'],
    sw[(262144,'in-place')]['frac_of_peak'], sw[(1048576,'in-place')]['frac_of_peak'], sw[(4194304,'in-place')]['frac_of_peak']))
ak=pol['f16x3']['actor_kernel']; ak32=pol['f32']['actor_kernel']
k5.append('| `k_actor_particle<4, f16x3>` / `<4, f32>` | 4096 envs × 4 agents = 16 384 rows | %s / %s µs | – | %.0f / %.0f TFLOP/s of network FLOPs = %.2f / %.2f of the f32 matrix peak | ~2 µs of every launch stage 60 KB of weights per workgroup |' % (
    f(ak['avg_launch_us'],2), f(ak32['avg_launch_us'],2), ak['roofline']['achieved'], ak32['roofline']['achieved'], ak['roofline']['frac'], ak32['roofline']['frac']))
table5='\n'.join(k5)
head=open(''+D+'/00_head.md').read()
sec45=open(''+D+'/45_sec45.md').read()
rep={'@@C5@@':f(b['c5']['us_per_tick'],2),'@@C5IN@@':f(b['c5']['launch_modes']['in_place_chains1']['us_per_tick'],2),'@@TABLE0@@':table0,'@@TABLE5@@':table5,'@@C2@@':f(b['c2']['us_per_tick'],2),'@@C3@@':f(b['c3']['us_per_tick'],2),'@@C4@@':f(b['c4']['us_per_tick'],2),
     '@@POL_EP@@':f(pol['f16x3']['one_launch_per_episode']['us_per_tick'],2),'@@POL_ACT@@':f(ak['avg_launch_us'],2),'@@POL_TICK@@':f(pol['f16x3']['launch_per_tick']['us_per_tick'],1)}
for k,v in rep.items():
    head=head.replace(k,v); sec45=sec45.replace(k,v)
out=head+open(''+D+'/10_sec1.md').read()+open(''+D+'/20_sec2.md').read()+open(''+D+'/30_sec3.md').read()+sec45+open(''+D+'/60_sec678.md').read()
open(ROOT+'/DESIGN.md','w').write(out)
print(len(out), out.count('@@'))
