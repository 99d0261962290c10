"""Beam bookkeeping of the narrator's `beam_sample` / `group_beam_search` (lavila/models/narrator.py:149-366).

The reference delegates it to `transformers.BeamSearchScorer` (pinned transformers==4.27: generation/beam_search.py).
This module keeps that class's observable behaviour -- which candidates become beams, when a hypothesis closes, when a
batch entry is finished, what `finalize` returns -- in a host-light form: per decoding step the candidate tensors are read
back ONCE (three small [entries, 2 * group] tensors) and the walk over them runs on Python lists; hypothesis token rows
stay on the device. Scores: a closed hypothesis of `n` tokens (the tokens before its eos) with summed log-probability `s`
scores `s / n ** length_penalty`; an entry keeps its `num_beams` best.
"""
import torch


class _Kept:
    """The up-to-`cap` best closed hypotheses of one batch entry, in insertion order."""

    __slots__ = ('cap', 'items', 'worst', 'count')

    def __init__(self, cap):
        self.cap, self.items, self.worst, self.count = cap, [], 1e9, 0

    def offer(self, score, tokens):
        if len(self.items) >= self.cap and not score > self.worst:
            return
        self.items.append((score, self.count, tokens))
        self.count += 1
        if len(self.items) > self.cap:
            self.items.remove(min(self.items, key=lambda t: (t[0], t[1])))      # lowest score, oldest first
            self.worst = min(t[0] for t in self.items)
        else:
            self.worst = min(score, self.worst)

    def best_first(self):
        """Highest score first; among equal scores the most recently closed one first (what a stable ascending sort
        popped from its end gives)."""
        return sorted(self.items, key=lambda t: (t[0], t[1]), reverse=True)


class BeamScorer:
    """entries = batch entries (clips x returned sequences for beam_sample, clips for group search); each entry runs
    `num_beams` beams in `num_beam_groups` groups of `group` beams; `process` is called once per group and step."""

    def __init__(self, entries, num_beams, device, length_penalty=1.0, early_stopping=False, keep=1, num_beam_groups=1):
        if not isinstance(num_beams, int) or num_beams <= 1:
            raise ValueError(f'`num_beams` has to be an integer strictly greater than 1, but is {num_beams}.')
        if not isinstance(num_beam_groups, int) or num_beam_groups > num_beams or num_beams % num_beam_groups != 0:
            raise ValueError('`num_beam_groups` has to be an integer smaller or equal than `num_beams` and `num_beams` has '
                             f'to be divisible by `num_beam_groups`, but is {num_beam_groups} with {num_beams}.')
        self.entries, self.num_beams, self.device = entries, num_beams, device
        self.group = num_beams // num_beam_groups
        self.length_penalty, self.early_stopping, self.keep = length_penalty, early_stopping, keep
        self.kept = [_Kept(num_beams) for _ in range(entries)]
        self.done = [False] * entries

    @property
    def is_done(self):
        return all(self.done)

    def process(self, input_ids, next_scores, next_tokens, next_indices, pad_token_id, eos_token_id):
        """input_ids [entries * group, cur_len]; candidates [entries, 2 * group] sorted by score (descending), `next_indices`
        = the beam (0 .. group-1 inside the entry) a candidate extends. Returns (scores, tokens, rows) of the beams of
        the next step, each [entries * group]; rows index input_ids."""
        cur_len = input_ids.shape[-1]
        if input_ids.shape[0] != self.entries * self.group:
            raise ValueError(f'A group beam size of {input_ids.shape[0]} is used as the input, but a group beam size of '
                             f'{self.group} is expected by the beam scorer.')
        sc, tk, ix = next_scores.tolist(), next_tokens.tolist(), next_indices.tolist()       # the step's one read-back
        out_s = [[0.0] * self.group for _ in range(self.entries)]
        out_t = [[0] * self.group for _ in range(self.entries)]
        out_r = [[0] * self.group for _ in range(self.entries)]
        for e in range(self.entries):
            if self.done[e]:
                if eos_token_id is None or pad_token_id is None:
                    raise ValueError('Generated beams >= num_beams -> eos_token_id and pad_token have to be defined')
                out_t[e] = [pad_token_id] * self.group
                continue
            filled = 0
            for rank, (tok, s, b) in enumerate(zip(tk[e], sc[e], ix[e])):
                row = e * self.group + b
                if eos_token_id is not None and tok == eos_token_id:
                    if rank >= self.group:               # an eos that is not among the entry's top `group` candidates
                        continue
                    n = input_ids.shape[-1]
                    self.kept[e].offer(s / (n ** self.length_penalty), input_ids[row].clone())
                else:
                    out_s[e][filled], out_t[e][filled], out_r[e][filled] = s, tok, row
                    filled += 1
                if filled == self.group:
                    break
            if filled < self.group:
                raise ValueError(f'At most {self.group} tokens in {tk[e]} can be equal to `eos_token_id: {eos_token_id}`.')
            k = self.kept[e]
            if len(k.items) >= k.cap and (self.early_stopping or k.worst >= max(sc[e]) / cur_len ** self.length_penalty):
                self.done[e] = True
        dev = input_ids.device
        return (torch.tensor(out_s, dtype=next_scores.dtype, device=dev).view(-1),
                torch.tensor(out_t, dtype=next_tokens.dtype, device=dev).view(-1),
                torch.tensor(out_r, dtype=next_indices.dtype, device=dev).view(-1))

    def finalize(self, input_ids, beam_scores, max_length, pad_token_id, eos_token_id):
        """input_ids [entries * num_beams, len], beam_scores [entries * num_beams]: the running beams of every unfinished
        entry close as they are; returns (sequences [entries * keep, <= max_length], scores [entries * keep])."""
        fs = beam_scores.tolist()
        n = input_ids.shape[-1]
        for e in range(self.entries):
            if self.done[e]:
                continue
            for b in range(self.num_beams):
                row = e * self.num_beams + b
                self.kept[e].offer(fs[row] / (n ** self.length_penalty), input_ids[row])
        best, scores = [], []
        for e in range(self.entries):
            ranked = self.kept[e].best_first()
            for j in range(self.keep):
                scores.append(ranked[j][0])
                best.append(ranked[j][2])
        lengths = [int(t.shape[-1]) for t in best]
        width = max(lengths) + 1
        if max_length is not None:
            width = min(width, max_length)
        seqs = input_ids.new_full((len(best), width), pad_token_id if pad_token_id is not None else 0)
        for i, t in enumerate(best):
            seqs[i, :lengths[i]] = t
            if lengths[i] < width:
                seqs[i, lengths[i]] = eos_token_id
        return seqs, torch.tensor(scores, dtype=torch.float32, device=input_ids.device)
