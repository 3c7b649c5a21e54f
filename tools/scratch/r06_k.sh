# under the 1400 W cap: what do the channel IQ's way to memory and back cost the streaming launch?  (timing-only variants)
bash tools/scratch/ab.sh 200 3 base small nostore nomem ring4 ring3
