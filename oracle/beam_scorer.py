"""TEST INFRASTRUCTURE ONLY -- restatement of `transformers.BeamSearchScorer` / `BeamHypotheses` as the reference uses
them (lavila/models/narrator.py:16,167-171,216-241,260-265,322-360).

PARITY UNPINNED against the dependency itself: the reference pins `transformers==4.27` (requirements.txt), that release is
not installed here (the image has 5.15, which no longer ships a BeamSearchScorer at all) and there is no network, so this
file restates the published algorithm of transformers 4.27 `generation/beam_search.py` (class BeamSearchScorer:
__init__ / is_done / process / finalize, class BeamHypotheses: add / is_done) and the parity of the product's beam
search is anchored on the REFERENCE'S OWN call sites: oracle/gen_golden.py runs the unmodified narrator.py
(`VCLM_HF.beam_sample`, `VCLM_HF.group_beam_search`) with this class installed under the name it imports.

Semantics kept from 4.27 (the points that decide outputs):
  * a hypothesis is (score = sum_logprobs / len(hyp) ** length_penalty, tokens); a batch entry keeps its `num_beams` best;
    `worst_score` tracks the lowest kept score;
  * process() walks the 2 * group_size candidates of a batch entry in order: an eos candidate among the first group_size
    ranks closes a hypothesis (tokens so far, without the eos), a later-ranked eos is skipped, every other candidate fills
    the next beam until group_size beams are set;
  * a batch entry is done when it holds num_beams hypotheses and (early stopping, or) the best running candidate cannot beat
    the worst kept one: worst_score >= best_sum_logprobs / cur_len ** length_penalty, with cur_len = the input length
    BEFORE the new token;
  * done entries are padded: scores 0, tokens pad, indices 0;
  * finalize() adds the running beams of entries that are not done, returns the `num_beam_hyps_to_keep` best per entry
    (best first), appends eos where it fits and pads with pad_token_id when lengths differ.
"""
from collections import UserDict

import torch


class BeamHypotheses:
    def __init__(self, num_beams, length_penalty, early_stopping, max_length=None):
        self.length_penalty = length_penalty
        self.early_stopping = early_stopping
        self.max_length = max_length
        self.num_beams = num_beams
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs, beam_indices=None):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp, beam_indices))
            if len(self) > self.num_beams:
                sorted_next_scores = sorted([(s, idx) for idx, (s, _, _) in enumerate(self.beams)])
                del self.beams[sorted_next_scores[0][1]]
                self.worst_score = sorted_next_scores[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping is True:
            return True
        highest_attainable_score = best_sum_logprobs / cur_len ** self.length_penalty
        return self.worst_score >= highest_attainable_score


class BeamSearchScorer:
    def __init__(self, batch_size, num_beams, device, length_penalty=1.0, do_early_stopping=False,
                 num_beam_hyps_to_keep=1, num_beam_groups=1, max_length=None):
        self.num_beams = num_beams
        self.device = device
        self.length_penalty = length_penalty
        self.do_early_stopping = do_early_stopping
        self.num_beam_hyps_to_keep = num_beam_hyps_to_keep
        self.num_beam_groups = num_beam_groups
        self.group_size = self.num_beams // self.num_beam_groups
        self._is_init = False
        self._beam_hyps = [BeamHypotheses(num_beams=self.num_beams, length_penalty=self.length_penalty,
                                          early_stopping=self.do_early_stopping, max_length=max_length)
                           for _ in range(batch_size)]
        self._done = torch.tensor([False for _ in range(batch_size)], dtype=torch.bool, device=self.device)
        if not isinstance(num_beams, int) or num_beams <= 1:
            raise ValueError(f"`num_beams` has to be an integer strictly greater than 1, but is {num_beams}.")
        if not isinstance(num_beam_groups, int) or (num_beam_groups > num_beams) or (num_beams % num_beam_groups != 0):
            raise ValueError("`num_beam_groups` has to be an integer smaller or equal than `num_beams` and `num_beams` "
                             f"has to be divisible by `num_beam_groups`, but is {num_beam_groups} with {num_beams}.")

    @property
    def is_done(self):
        return self._done.all()

    def process(self, input_ids, next_scores, next_tokens, next_indices, pad_token_id=None, eos_token_id=None,
                beam_indices=None):
        cur_len = input_ids.shape[-1]
        batch_size = len(self._beam_hyps)
        if not (batch_size == (input_ids.shape[0] // self.group_size)):
            raise ValueError(f"A group beam size of {input_ids.shape[0]} is used as the input, but a group beam size of "
                             f"{self.group_size} is expected by the beam scorer.")
        device = input_ids.device
        next_beam_scores = torch.zeros((batch_size, self.group_size), dtype=next_scores.dtype, device=device)
        next_beam_tokens = torch.zeros((batch_size, self.group_size), dtype=next_tokens.dtype, device=device)
        next_beam_indices = torch.zeros((batch_size, self.group_size), dtype=next_indices.dtype, device=device)
        if isinstance(eos_token_id, int):
            eos_token_id = [eos_token_id]

        for batch_idx, beam_hyp in enumerate(self._beam_hyps):
            if self._done[batch_idx]:
                if self.num_beams < len(beam_hyp):
                    raise ValueError(f"Batch can only be done if at least {self.num_beams} beams have been generated")
                if eos_token_id is None or pad_token_id is None:
                    raise ValueError("Generated beams >= num_beams -> eos_token_id and pad_token have to be defined")
                next_beam_scores[batch_idx, :] = 0
                next_beam_tokens[batch_idx, :] = pad_token_id
                next_beam_indices[batch_idx, :] = 0
                continue

            beam_idx = 0
            for beam_token_rank, (next_token, next_score, next_index) in enumerate(
                    zip(next_tokens[batch_idx], next_scores[batch_idx], next_indices[batch_idx])):
                batch_beam_idx = batch_idx * self.group_size + next_index
                if (eos_token_id is not None) and (next_token.item() in eos_token_id):
                    if beam_token_rank >= self.group_size:
                        continue
                    beam_hyp.add(input_ids[batch_beam_idx].clone(), next_score.item(), beam_indices=None)
                else:
                    next_beam_scores[batch_idx, beam_idx] = next_score
                    next_beam_tokens[batch_idx, beam_idx] = next_token
                    next_beam_indices[batch_idx, beam_idx] = batch_beam_idx
                    beam_idx += 1
                if beam_idx == self.group_size:
                    break

            if beam_idx < self.group_size:
                raise ValueError(f"At most {self.group_size} tokens in {next_tokens[batch_idx]} can be equal to "
                                 f"`eos_token_id: {eos_token_id}`. Make sure {next_tokens[batch_idx]} are corrected.")
            self._done[batch_idx] = self._done[batch_idx] or beam_hyp.is_done(next_scores[batch_idx].max().item(), cur_len)

        return UserDict({"next_beam_scores": next_beam_scores.view(-1), "next_beam_tokens": next_beam_tokens.view(-1),
                         "next_beam_indices": next_beam_indices.view(-1)})

    def finalize(self, input_ids, final_beam_scores, final_beam_tokens, final_beam_indices, max_length,
                 pad_token_id=None, eos_token_id=None, beam_indices=None):
        batch_size = len(self._beam_hyps)
        if isinstance(eos_token_id, int):
            eos_token_id = [eos_token_id]

        for batch_idx, beam_hyp in enumerate(self._beam_hyps):
            if self._done[batch_idx]:
                continue
            for beam_id in range(self.num_beams):
                batch_beam_idx = batch_idx * self.num_beams + beam_id
                final_score = final_beam_scores[batch_beam_idx].item()
                final_tokens = input_ids[batch_beam_idx]
                beam_hyp.add(final_tokens, final_score, beam_indices=None)

        sent_lengths = input_ids.new(batch_size * self.num_beam_hyps_to_keep)
        best = []
        best_scores = torch.zeros(batch_size * self.num_beam_hyps_to_keep, device=self.device, dtype=torch.float32)
        for i, beam_hyp in enumerate(self._beam_hyps):
            sorted_hyps = sorted(beam_hyp.beams, key=lambda x: x[0])
            for j in range(self.num_beam_hyps_to_keep):
                best_hyp_tuple = sorted_hyps.pop()
                best_score, best_hyp = best_hyp_tuple[0], best_hyp_tuple[1]
                sent_lengths[self.num_beam_hyps_to_keep * i + j] = len(best_hyp)
                best.append(best_hyp)
                best_scores[i * self.num_beam_hyps_to_keep + j] = best_score

        sent_lengths_max = sent_lengths.max().item() + 1
        sent_max_len = min(sent_lengths_max, max_length) if max_length is not None else sent_lengths_max
        decoded = input_ids.new(batch_size * self.num_beam_hyps_to_keep, sent_max_len)
        if sent_lengths.min().item() != sent_lengths.max().item():
            assert pad_token_id is not None, "`pad_token_id` has to be defined"
            decoded.fill_(pad_token_id)
        for i, hypo in enumerate(best):
            decoded[i, : sent_lengths[i]] = hypo
            if sent_lengths[i] < sent_max_len:
                decoded[i, sent_lengths[i]] = eos_token_id[0]
        return UserDict({"sequences": decoded, "sequence_scores": best_scores, "beam_indices": None})
