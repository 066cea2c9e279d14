"""Prompt builder and generate_long orchestration against fixtures produced by the
UNMODIFIED reference classes (tests/golden/prompt_cases.json <- oracle/gen_golden_prompt.py): same tokenizer stand-in
(oracle/fake_tokenizer.ByteTokenizer), same stubbed `generate`, so every prompt that reaches the model and every
code block that comes back out must be identical."""
import json
import os
import queue

import pytest
import torch

from fish_speech_amd import text2semantic as T2S
from fish_speech_amd.prompt import Conversation, Message, TextPart, VQPart
from oracle.fake_tokenizer import ByteTokenizer
from oracle.gen_golden_prompt import NCB, codes_for, stub_generate_factory

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prompt_cases.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def _parts(spec):
    out = []
    for p in spec:
        if p[0] == "text":
            out.append(TextPart(text=p[1]))
        elif p[0] == "tokens":
            out.append(TextPart(tokens=list(p[1])))
        else:
            out.append(VQPart(codes=codes_for(p[1], p[2])))
    return out


def test_prompts_equal_the_reference_conversation_encoding(gold):
    tok = ByteTokenizer()
    assert gold["num_codebooks"] == NCB
    for case in gold["conversations"]:
        conv = Conversation()
        for m in case["messages"]:
            kw = {k: v for k, v in m.items() if k not in ("role", "parts")}
            conv.append(Message(role=m["role"], parts=_parts(m["parts"]), **kw))
        values, masks, parts = conv.encode_for_inference(tok, num_codebooks=NCB)
        want = torch.tensor(case["values"], dtype=torch.int64).reshape(NCB + 1, -1)
        assert masks is None and parts is None
        assert values.dtype == torch.int64 and values.shape == want.shape, case["name"]
        assert torch.equal(values, want), case["name"]


def test_prompt_layout_properties():
    """Independent of the fixture: VQ columns carry semantic ids in row 0 and the codes below, text columns carry
    zeros below; a VQ part with the wrong codebook count is rejected; a TextPart needs text or tokens."""
    tok = ByteTokenizer()
    codes = codes_for(3, 6)
    conv = Conversation([Message(role="user", parts=[TextPart(text="ab"), VQPart(codes=codes)])])
    v, _, _ = conv.encode_for_inference(tok, NCB)
    vq = v[0] >= tok.semantic_begin_id
    assert int(vq.sum()) == 6
    assert torch.equal(v[1:, vq], codes) and torch.equal(v[0, vq], codes[0] + tok.semantic_begin_id)
    assert int(v[1:, ~vq].abs().sum()) == 0
    with pytest.raises(ValueError):
        Conversation([Message(role="user", parts=[VQPart(codes=codes[:4])])]).encode_for_inference(tok, NCB)
    with pytest.raises(ValueError):
        TextPart()


def test_turn_splitting_and_batching_equal_the_reference(gold):
    for c in gold["split"]:
        turns = T2S.split_text_by_speaker(c["text"])
        assert turns == c["turns"], c["text"]
        assert T2S.group_turns_into_batches(turns, max_speakers=c["max_speakers"], max_bytes=c["max_bytes"]) == c["batches"]


class _Cfg:
    num_codebooks = NCB
    max_seq_len = 4096


class _FakeModel:
    def __init__(self, max_seq_len=4096):
        self.config = _Cfg()
        self.config.max_seq_len = max_seq_len
        self.tokenizer = ByteTokenizer()


def test_generate_long_call_trace_equals_the_reference(gold, monkeypatch):
    for entry in gold["generate_long"]:
        case = entry["case"]
        log = []
        model = _FakeModel()
        monkeypatch.setattr(T2S, "generate", stub_generate_factory(model.tokenizer, log))
        kw = {k: v for k, v in case.items() if k not in ("name", "prompt_vq")}
        if "prompt_vq" in case:
            kw["prompt_tokens"] = [codes_for(s, n) for s, n in case["prompt_vq"]]
        got = list(T2S.generate_long(model=model, device="cpu", decode_one_token=None, **kw))
        assert log == entry["calls"], case["name"]                      # identical prompts + sampling parameters
        assert len(got) == len(entry["responses"]), case["name"]
        for g, w in zip(got, entry["responses"]):
            assert g.action == w["action"] and g.text == w["text"]
            if w["codes"] is None:
                assert g.codes is None
            else:
                assert g.codes.cpu().numpy().tolist() == w["codes"]


def test_generate_long_rejects_long_prompts_like_the_reference(gold, monkeypatch):
    model = _FakeModel(max_seq_len=2048 + 40)
    monkeypatch.setattr(T2S, "generate", stub_generate_factory(model.tokenizer, []))
    with pytest.raises(ValueError) as e:
        list(T2S.generate_long(model=model, device="cpu", text="<|speaker:0|>" + "w" * 64))
    assert str(e.value) == gold["too_long_error"]
    for bad in (dict(top_p=0.0), dict(temperature=2.0)):
        with pytest.raises(AssertionError):
            list(T2S.generate_long(model=model, device="cpu", text="x", **bad))


def test_worker_queue_streams_responses_and_wraps_errors(monkeypatch):
    class M:
        config = _Cfg()

        def setup_caches(self, **kw):
            self.caches = kw

        def parameters(self):
            return [torch.zeros(1, dtype=torch.bfloat16)]

    m = M()
    monkeypatch.setattr(T2S, "init_model", lambda *a, **k: (m, "decode-fn"))

    def fake_long(*, model, decode_one_token, text, **kw):
        assert model is m and decode_one_token == "decode-fn"
        if text == "boom":
            raise RuntimeError("kernel fault")
        yield T2S.GenerateResponse(action="sample", codes=torch.ones(NCB, 2, dtype=torch.int64), text=text)
        yield T2S.GenerateResponse(action="next")

    monkeypatch.setattr(T2S, "generate_long", fake_long)
    q_in = T2S.launch_thread_safe_queue("ckpt", "cuda:0", torch.bfloat16)
    assert m.caches["max_batch_size"] == 1 and m.caches["max_seq_len"] == _Cfg.max_seq_len
    out = queue.Queue()
    q_in.put(T2S.GenerateRequest(request=dict(text="hi", device="cpu"), response_queue=out))
    a, b = out.get(timeout=10), out.get(timeout=10)
    assert a.status == "success" and a.response.action == "sample" and a.response.text == "hi"
    assert b.status == "success" and b.response.action == "next"
    q_in.put(T2S.GenerateRequest(request=dict(text="boom", device="cpu"), response_queue=out))
    e = out.get(timeout=10)
    assert e.status == "error" and isinstance(e.response, RuntimeError)
    q_in.put(T2S.GenerateRequest(request=dict(text="again", device="cpu"), response_queue=out))   # still alive
    assert out.get(timeout=10).response.text == "again"
    q_in.put(None)

    def failing_init(*a, **k):
        raise FileNotFoundError("no checkpoint")

    monkeypatch.setattr(T2S, "init_model", failing_init)
    with pytest.raises(FileNotFoundError):
        T2S.launch_thread_safe_queue("missing", "cuda:0", torch.bfloat16)


def test_cli_keeps_the_reference_flags(tmp_path):
    from click.testing import CliRunner

    main = T2S._cli()
    res = CliRunner().invoke(main, ["--help"])
    assert res.exit_code == 0
    for flag in ("--text", "--prompt-text", "--prompt-tokens", "--prompt-audio", "--output", "--num-samples",
                 "--max-new-tokens", "--top-p", "--top-k", "--temperature", "--checkpoint-path", "--device",
                 "--compile", "--no-compile", "--seed", "--half", "--iterative-prompt", "--chunk-length", "--output-dir"):
        assert flag in res.output, flag
    ck = tmp_path / "ckpt"
    ck.mkdir()
    res = CliRunner().invoke(main, ["--checkpoint-path", str(ck), "--prompt-text", "x", "--output-dir", str(tmp_path / "o")])
    assert isinstance(res.exception, ValueError) and "--prompt-text requires" in str(res.exception)
    res = CliRunner().invoke(main, ["--checkpoint-path", str(ck), "--half", "--output-dir", str(tmp_path / "o")])
    assert res.exit_code != 0 and "bf16" in res.output
