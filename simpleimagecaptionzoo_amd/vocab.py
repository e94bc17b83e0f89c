"""Vocabulary with the interface of the reference's Caption_Vocabulary (ClassRepository/CaptionVocabClass.py): callable
word -> id with the <unk> fallback, `word2ix`, `ix2word`, `add_word`, `len()`.  Ids are dense and assigned in insertion order;
the reference's builder inserts the specials first, so ids 0..3 are <pad>, <sta>, <end>, <unk>
(PreProcess/Build_caption_vocab.py:37-40) -- the decoders rely on <sta> = 1 and <end> = 2."""
from collections.abc import Mapping

SPECIALS = ("<pad>", "<sta>", "<end>", "<unk>")


class _WordsById(Mapping):
    """Read-only id -> word view over the word list (what the reference keeps as a second dict)."""

    def __init__(self, words):
        self._words = words

    def __getitem__(self, i):
        i = int(i)
        if i < 0 or i >= len(self._words):
            raise KeyError(i)
        return self._words[i]

    def __iter__(self):
        return iter(range(len(self._words)))

    def __len__(self):
        return len(self._words)


class Caption_Vocabulary:
    def __init__(self, words=()):
        self._words = []
        self.word2ix = {}
        for w in words:
            self.add_word(w)

    @classmethod
    def from_reference(cls, obj):
        """Adopt an unpickled reference vocabulary (it carries `ix2word` as {id: word})."""
        return cls(obj.ix2word[i] for i in range(len(obj.ix2word)))

    @property
    def ix2word(self):
        return _WordsById(self._words)

    @property
    def idx(self):
        return len(self._words)

    def add_word(self, new_word):
        if new_word in self.word2ix:
            return self.word2ix[new_word]
        self.word2ix[new_word] = len(self._words)
        self._words.append(new_word)
        return len(self._words) - 1

    def __call__(self, word):
        i = self.word2ix.get(word)
        return self.word2ix["<unk>"] if i is None else i

    def __len__(self):
        return len(self._words)


def synthetic_vocab(V):
    return Caption_Vocabulary(SPECIALS + tuple("w%d" % i for i in range(V - len(SPECIALS))))
