"""The recogniser's character set: the first line of the reference's `data/alphabet.txt` (data, read by
`src/utils.py:26-27`); class 0 is the CTC blank, character i has class i + 1."""
ALPHABET = '7BCNTh2!F\'P0ouRvz3[Qdesr6#:ÉyU(4bt%"?´Kl.ZOM8@A1+)/ ki&DW$fwn;=p5HqSjV]JX-GEagxILmYc9,'
