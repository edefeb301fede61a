import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print(sys.argv[1], 'value', j['value'], 'dominant', r['achieved'], r['avg_launch_us'])
for e in r['gemm_shapes'][:9]: print('   ',e['M'],e['N'],e['K'],e['out'],'res' if e['residual'] else '', e['act'], e['launches'], e['avg_launch_us'],e['tflops'])
