import json,sys
for f in sys.argv[1:]:
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    r=d['roofline']; ss=r.get('steady_state',{})
    print(f, 'value', d['value'], d['value_min_max'], 'run', r['kernel'], 'frac', r['frac'], 'of', r['overfetch'], 'valu', r['valu_frac'])
    print('   ss', ss.get('avg_launch_ms'), ss.get('value'), 'frac', ss.get('frac'), 'of', ss.get('overfetch'), 'valu', ss.get('valu_frac'), 'rw', ss.get('traffic_read_write'), ss.get('launch_shape'))
    for k,v in ss.get('clock_vs_launch',{}).items(): print('   ', k, v['mhz_ms_per_8_launch_group'][-3:], v['value'])
    for s in d.get('secondary',[]):
        if 'error' in s: print('   sec', s); continue
        rr=s['roofline']; s2=rr.get('steady_state',{})
        print('   sec', ('FUSED ' if 'arithmetic' in s else '') + s['config']['workload'][:44], s['dtype'], s['value'], 'frac', rr['frac'], 'of', rr['overfetch'], 'ss', s2.get('value'), s2.get('avg_launch_ms'))
