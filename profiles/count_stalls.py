import sys,re,ast
rows=[l for l in sys.stdin if ' call ' in l]
bad=0; tot=0; times=[]
for l in rows[3:]:
    d=ast.literal_eval(l[l.index('{'):])
    ms=float(re.search(r': ([0-9.]+) ms',l).group(1)); times.append(ms)
    tot+=1
    if d['vertices']>8 or d['vertex_caches']>8 or d['connect']>60: bad+=1
times.sort()
print("calls %d, with a stalled phase %d, median %.1f ms, min %.1f, max %.1f" % (tot,bad,times[len(times)//2],times[0],times[-1]))
