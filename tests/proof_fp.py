"""Fingerprint of a PRODUCT proof (ligero_amd.prover.Proof) in the form tests/golden/proofs.json records: SHA-256 and byte length of
each of the ten fields lgp_proof_field_bytes exports.  Uses hashlib only (no oracle: spawned rank workers import this too)."""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def fingerprint(proof):
    fb = proof.field_bytes()
    fp = {name: hashlib.sha256(b).hexdigest() for name, b in fb.items()}
    fp["lens"] = {name: len(b) for name, b in fb.items()}
    return fp


def golden():
    return json.load(open(os.path.join(HERE, "golden", "proofs.json")))


def same(fp, want):
    """fp == the golden entry, field by field (the golden entry also carries dims / accepted)"""
    return all(fp[f] == want[f] for f in fp if f != "lens") and fp["lens"] == want["lens"]


def diff(fp, want):
    return [f for f in fp if f != "lens" and fp[f] != want[f]] + [f"len({f})" for f in fp["lens"] if fp["lens"][f] != want["lens"][f]]
