"""Test-only engine: same interface as squarna_amd.engine.HipEngine, answers come from the
CPU oracle.  Installed with squarna_amd.engine.use_engine() by CPU tests to check the
host-side text layer (parsers, Predict, RunSQRNdbnseq printing) against the golden texts.
The product never imports this."""
from oracle import sqrn_oracle as O


class OracleEngine:
    name = "oracle"

    def fold_records(self, records, **opts):
        opts.pop("keep", None)                   # (an engine option: how many structures to fetch; the oracle returns all)
        out = []
        for rec in records:
            seq, reacts, restraints, dbn, paramsets = rec[:5]
            sm = rec[5] if len(rec) > 5 else None
            out.append(O.SQRNdbnseq(seq, reacts, restraints, dbn, paramsets, stemmatrix=sm, **opts))
        return out

    def entropy(self, record, interchainonly=False):
        seq, reacts, restraints, dbn, paramsets = record[:5]
        sm = record[5] if len(record) > 5 else None
        return O.SQRNdbnseq(seq, reacts, restraints, dbn, paramsets, entropy=True, stemmatrix=sm,
                            interchainonly=interchainonly)

    def yield_stems(self, records, bpweights, minlen, minbpscore, interchainonly=False):
        out = []
        for seq, reacts, restraints in records:
            seq = seq.upper().replace("T", "U")
            restraints = restraints or "." * len(seq)
            shortseq, shortrest = O.UnAlign(seq, restraints)
            rc = [reacts[i] for i in range(len(seq)) if seq[i] not in O.GAPS] if reacts else None
            rbps, rxs, rl, rr = O.ParseRestraints(shortrest)
            b, s = O.BPMatrix(shortseq, bpweights, rxs, rl, rr, interchainonly, rc)
            out.append((shortseq, O.AnnotateStems(b, s, rbps, [], minlen, minbpscore)))
        return out
